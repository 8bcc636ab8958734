"""Pins against the third-party packages the reference calls on this path, generated with the REAL packages
(oracle/make_golden_thirdparty.py under /opt/conda/bin/python3.9 -> tests/golden/thirdparty_pins.npz):
PyWavelets 1.1.1 (condition/utils.py:106-139), scikit-image 0.18.3 (sample_condition_openai.py:44-45), SciPy 1.7.1 with the legacy
`cg(tol=)` keyword (condition/condition.py:343,379,432).  CPU: the oracle restatements and the host metrics; GPU: the HIP DWT kernels."""
import numpy as np
import pytest
import torch

from oracle import transforms as otf


@pytest.fixture(scope="module")
def pins(gold):
    return gold("thirdparty_pins")


def _dwt_inputs(pins, tag):
    if tag == "f32_256":          # the full-size plane's inputs are regenerated, not stored (oracle/make_golden_thirdparty.py)
        r = np.random.RandomState(256)
        return torch.from_numpy(r.randn(1, 1, 256, 256).astype(np.float32)), torch.from_numpy(r.randn(1, 1, 256, 256).astype(np.float32))
    return torch.from_numpy(pins[f"dwt_x_{tag}"]), torch.from_numpy(pins[f"dwt_c_{tag}"])


@pytest.mark.parametrize("tag,tol", [("f32_64", 2e-6), ("f64_64", 1e-13), ("f32_256", 2e-6)])
def test_oracle_dwt_matches_pywt(pins, tag, tol):
    """wavedec2('haar', level 3) + coeffs_to_array and array_to_coeffs + waverec2 of PyWavelets vs oracle.transforms (layout AND signs)."""
    x, c = _dwt_inputs(pins, tag)
    assert float((otf.dwt_haar(x) - torch.from_numpy(pins[f"dwt_fwd_{tag}"])).abs().max()) < tol
    assert float((otf.idwt_haar(c) - torch.from_numpy(pins[f"dwt_inv_{tag}"])).abs().max()) < tol


def test_oracle_dwt_subband_placement_matches_pywt(pins):
    """unit impulses in cA3 / top-right / bottom-left / bottom-right level-1 blocks of an 8 x 8 coefficient array -> pywt's images:
    top-right varies along the COLUMNS ('ad'), bottom-left along the ROWS ('da')."""
    x = otf.idwt_haar(torch.from_numpy(pins["dwt_impulse_c"]))
    assert float((x - torch.from_numpy(pins["dwt_impulse_x"])).abs().max()) < 1e-14
    px = pins["dwt_impulse_x"]
    assert px[1, 0, 0] == -px[1, 0, 1] == px[1, 1, 0] and px[2, 0, 0] == px[2, 0, 1] == -px[2, 1, 0]


def test_host_metrics_match_skimage(pins):
    """kdip_amd.metrics PSNR / SSIM (what compute_metrics reports) vs skimage.metrics.peak_signal_noise_ratio / structural_similarity
    (multichannel, data_range 1) on four [3, H, W] pairs incl. a non-square one."""
    import kdip_amd.metrics as km
    for k in range(int(pins["n_img"])):
        a, b = torch.from_numpy(pins[f"img_a_{k}"]), torch.from_numpy(pins[f"img_b_{k}"])
        assert abs(km.peak_signal_noise_ratio(a, b, 1.0) - float(pins[f"psnr_{k}"])) < 1e-6
        assert abs(km.structural_similarity(a, b, 1.0) - float(pins[f"ssim_{k}"])) < 1e-10


def test_oracle_cg_matches_legacy_scipy(pins):
    """scipy 1.7.1 `cg(A, b, tol=1e-4, maxiter=1000)` (the legacy keyword the reference uses) on a float32 SPD system of the mat-solver's
    form: exit code 0, residual below tol * ||b||; the oracle's restated CG and the `rtol=tol, atol=0` rebinding of oracle/refimport.py on
    today's scipy stop within 2 iterations of it (fp32 recurrences: the iteration at which ||r|| crosses the threshold moves by one or two
    with the summation order) and agree with its solution to 1e-4 relative -- the tolerance itself."""
    import scipy.sparse.linalg as spla
    from oracle.solvers import cg_batched
    A, b, x_ref = pins["cg_A"], pins["cg_b"], pins["cg_x"]
    n_ref = int(pins["cg_iters"])
    assert int(pins["cg_info"]) == 0 and float(pins["cg_rel_residual"]) < 1e-4 * 1.01
    At = torch.from_numpy(A)
    x, it, info = cg_batched(lambda p: p @ At.T, torch.from_numpy(b)[None], tol=1e-4, maxiter=1000)
    scale = float(np.abs(x_ref).max())
    assert int(info[0]) == 0 and abs(int(it[0]) - n_ref) <= 2, (int(it[0]), n_ref)
    assert float((x[0] - torch.from_numpy(x_ref)).abs().max()) < 1e-4 * scale
    its = [0]
    xs, inf = spla.cg(A, b, rtol=1e-4, atol=0.0, maxiter=1000, callback=lambda xk: its.__setitem__(0, its[0] + 1))
    assert inf == 0 and abs(its[0] - n_ref) <= 2 and float(np.abs(xs - x_ref).max()) < 1e-4 * scale
    # the well-conditioned system (the mat-solver's regime): identical iteration count, both ways
    A2 = A + (float(pins["cg2_A_diag_shift"]) - 0.05 ** 2) * np.eye(A.shape[0], dtype=np.float32)
    n2 = int(pins["cg2_iters"])
    A2t = torch.from_numpy(A2)
    x2, it2, info2 = cg_batched(lambda p: p @ A2t.T, torch.from_numpy(b)[None], tol=1e-4, maxiter=1000)
    assert int(pins["cg2_info"]) == 0 and int(info2[0]) == 0 and int(it2[0]) == n2, (int(it2[0]), n2)
    assert float((x2[0] - torch.from_numpy(pins["cg2_x"])).abs().max()) < 1e-4 * float(np.abs(pins["cg2_x"]).max())
    its[0] = 0
    xs2, inf2 = spla.cg(A2, b, rtol=1e-4, atol=0.0, maxiter=1000, callback=lambda xk: its.__setitem__(0, its[0] + 1))
    assert inf2 == 0 and its[0] == n2


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["f32_64", "f32_256"])
def test_hip_dwt_matches_pywt(pins, tag):
    """The HIP Haar-3 kernels (kdip_op_ortho: dwt_haar3 / idwt_haar3) against PyWavelets' own output, 64 x 64 and the full 256 x 256."""
    from kdip_amd.transforms import OrthoTransform
    ot = OrthoTransform("dwt")
    x, c = _dwt_inputs(pins, tag)
    fwd, inv = torch.from_numpy(pins[f"dwt_fwd_{tag}"]), torch.from_numpy(pins[f"dwt_inv_{tag}"])
    if x.shape[1] == 1:          # the operator context works on 3-channel images: the transform is per plane
        x, c, fwd, inv = (t.repeat(1, 3, 1, 1) for t in (x, c, fwd, inv))
    assert float((ot(x.cuda()).cpu() - fwd).abs().max()) < 2e-6
    assert float((ot.inv(c.cuda()).cpu() - inv).abs().max()) < 2e-6
