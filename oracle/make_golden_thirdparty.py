"""Golden vectors from the THIRD-PARTY packages the reference calls on this path, generated with the real packages.

TEST INFRASTRUCTURE.  Run in the build container with the conda interpreter that ships them (found in round 5: the system Python 3.10
of this image has neither, /opt/conda/bin/python3.9 has both -- numpy only there, no torch):

    /opt/conda/bin/python3.9 oracle/make_golden_thirdparty.py        ->  tests/golden/thirdparty_pins.npz

  * PyWavelets 1.1.1  -- condition/utils.py:106-139: `pywt.wavedec2(x, 'haar', level=3, axes=(-2, -1))` + `pywt.coeffs_to_array`,
    and the inverse `pywt.array_to_coeffs(..., output_format='wavedec2')` + `pywt.waverec2`: the Haar-3 Mallat layout that
    oracle/transforms.py and the HIP kernels (`dwt_haar3` / `idwt_haar3`) restate.  (The reference's environment.yml does not pin
    PyWavelets; 1.1.1 is what is installable here.  The Haar filters and the coeffs_to_array layout have not changed across 1.x.)
  * scikit-image 0.18.3 -- sample_condition_openai.py:44-45: `peak_signal_noise_ratio(a, b, data_range=1)` and
    `structural_similarity(a, b, channel_axis=0, data_range=1)`.  0.18 spells the channel argument `multichannel=True` (last axis):
    the same algorithm (per-channel SSIM with the 7x7 uniform window, sample covariance, cropped mean; mean over channels) --
    `channel_axis` only arrived in 0.19.
  * SciPy 1.7.1 -- condition/condition.py:343,379,432: `scipy.sparse.linalg.cg(A, b, tol=1e-4, maxiter=1000)` with the LEGACY `tol`
    keyword the reference was written against (removed in SciPy 1.14; oracle/refimport.py rebinds it to `rtol=tol, atol=0`): solution,
    iteration count and exit code of that call on a seeded SPD system, to pin the restated stopping rule ||r|| <= tol * ||b||.

Only inputs and expected outputs are stored."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    import pywt
    import scipy
    import scipy.sparse.linalg as spla
    import skimage
    from skimage.metrics import peak_signal_noise_ratio, structural_similarity

    out = {"versions": np.array([f"pywt {pywt.__version__}", f"skimage {skimage.__version__}", f"scipy {scipy.__version__}", f"numpy {np.__version__}"])}
    rng = np.random.RandomState(0)

    # ---- PyWavelets: forward (wavedec2 + coeffs_to_array) and inverse (array_to_coeffs + waverec2), fp32 and fp64, two sizes
    for tag, shape, dt in (("f32_64", (2, 3, 64, 64), np.float32), ("f64_64", (1, 3, 64, 64), np.float64), ("f32_256", (1, 1, 256, 256), np.float32)):
        # the full-size plane: inputs are NOT stored, the tests regenerate them (np.random.RandomState(256): the legacy generator is
        # stable across numpy versions); only pywt's outputs are
        r = np.random.RandomState(256) if tag == "f32_256" else rng
        x = r.randn(*shape).astype(dt)
        co = pywt.wavedec2(x, wavelet="haar", level=3, axes=(-2, -1))
        arr, sl = pywt.coeffs_to_array(co, axes=(-2, -1))
        c = r.randn(*shape).astype(dt)
        rec = pywt.waverec2(pywt.array_to_coeffs(c, sl, output_format="wavedec2"), wavelet="haar", axes=(-2, -1))
        assert arr.shape == x.shape and rec.shape == x.shape and arr.dtype == dt
        out[f"dwt_fwd_{tag}"], out[f"dwt_inv_{tag}"] = arr, rec.astype(dt)
        if tag != "f32_256":
            out[f"dwt_x_{tag}"], out[f"dwt_c_{tag}"] = x, c
    # sub-band placement, stated explicitly: a unit impulse in each level-1 sub-band of an 8x8 coefficient array -> image (level 3 of an 8x8 = 1x1 blocks)
    sl8 = pywt.coeffs_to_array(pywt.wavedec2(np.zeros((8, 8)), "haar", level=3), axes=(-2, -1))[1]
    imp = np.zeros((4, 8, 8))
    for k, (r, c_) in enumerate(((0, 0), (0, 4), (4, 0), (4, 4))):       # cA3 | level-1 top-right | bottom-left | bottom-right blocks
        imp[k, r, c_] = 1.0
    out["dwt_impulse_c"] = imp
    out["dwt_impulse_x"] = np.stack([pywt.waverec2(pywt.array_to_coeffs(imp[k], sl8, output_format="wavedec2"), "haar") for k in range(4)])

    # ---- scikit-image: PSNR / SSIM of [0, 1] image pairs [3, H, W] as compute_metrics builds them
    A, Bm = [], []
    for k, (h, w, noise) in enumerate(((64, 64, 0.05), (64, 64, 0.3), (96, 80, 0.1), (128, 128, 0.02))):
        base = rng.rand(3, h, w)
        base = (base + np.roll(base, 1, 1) + np.roll(base, 1, 2) + np.roll(base, 2, 1)) / 4       # some spatial structure
        a = np.clip(base, 0, 1).astype(np.float32)
        b = np.clip(base + noise * rng.randn(3, h, w), 0, 1).astype(np.float32)
        ps = peak_signal_noise_ratio(a, b, data_range=1)
        ss = structural_similarity(np.moveaxis(a, 0, -1), np.moveaxis(b, 0, -1), multichannel=True, data_range=1)
        out[f"img_a_{k}"], out[f"img_b_{k}"], out[f"psnr_{k}"], out[f"ssim_{k}"] = a, b, np.float64(ps), np.float64(ss)
    out["n_img"] = np.int64(4)

    # ---- SciPy legacy cg(tol=): float32 SPD system of the mat-solver's shape  sigma^2 I + K diag(v) K^T
    n = 240
    K = rng.randn(n, n).astype(np.float32) / np.sqrt(n)
    v = (0.05 + rng.rand(n)).astype(np.float32)
    Amat = (0.05 ** 2) * np.eye(n, dtype=np.float32) + (K * v) @ K.T
    bvec = rng.randn(n).astype(np.float32)
    its = [0]

    def cb(xk):
        its[0] += 1
    xs, info = spla.cg(Amat, bvec, tol=1e-4, maxiter=1000, callback=cb)
    out["cg_A"], out["cg_b"], out["cg_x"], out["cg_iters"], out["cg_info"] = Amat, bvec, xs.astype(np.float32), np.int64(its[0]), np.int64(info)
    res = np.linalg.norm(bvec - Amat @ xs) / np.linalg.norm(bvec)
    out["cg_rel_residual"] = np.float64(res)

    # ... and a well-conditioned one (the mat-solver's regime: 3 - 11 iterations): the iteration count must match exactly
    A2 = 0.5 * np.eye(n, dtype=np.float32) + (K * v) @ K.T
    its[0] = 0
    x2, info2 = spla.cg(A2, bvec, tol=1e-4, maxiter=1000, callback=cb)
    out["cg2_A_diag_shift"], out["cg2_x"], out["cg2_iters"], out["cg2_info"] = np.float32(0.5), x2.astype(np.float32), np.int64(its[0]), np.int64(info2)

    path = os.path.join(ROOT, "tests", "golden", "thirdparty_pins.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.startswith(("cg_i", "cg_rel", "psnr", "ssim", "versions"))})


if __name__ == "__main__":
    sys.exit(main())
