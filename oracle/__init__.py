"""CPU oracle for the guided-diffusion inverse-problem sampler hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

This package is a from-scratch CPU restatement (torch-CPU fp32 / numpy) of the
reference algorithm on the path named by BASELINE.json `north_star`
(SURVEY.md section 8a rows A1-A16).  Every function cites the reference
file:line it follows.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it, and only as the checker /
the timed CPU baseline -- the product package
(`k-diffusion-inverse-problems_amd/`) never imports it and fails loudly if the
HIP library is missing.

Why torch-CPU rather than plain C/numpy: the path is a floating-point UNet
(388 GFLOP forward per image); the reference itself *is* torch-CPU fp32
(`use_fp16=False`, condition/diffpir_utils/utils_model.py:364), so conv /
group-norm / softmax here call the same ATen CPU ops and autograd provides the
VJP exactly as the reference does (condition/condition.py:172).

Parity pinning: the restatement is validated against the real reference,
imported in the build container by `oracle/make_golden.py` (which also writes
the committed fixtures under tests/golden/).  Third-party boundaries
(SURVEY.md section 8c): PyWavelets, scikit-image and the legacy-`tol` SciPy `cg`
are pinned against the real packages (found in /opt/conda/bin/python3.9:
`oracle/make_golden_thirdparty.py` -> tests/golden/thirdparty_pins.npz);
GPyTorch (`autoI`, restated as Type-I with a CG solve) and `lpips` cannot be
executed here and stay **parity unpinned**.
"""
