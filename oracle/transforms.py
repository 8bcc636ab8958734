"""Oracle: orthogonal basis transforms (condition/utils.py:50-139).

TEST INFRASTRUCTURE (see oracle/__init__.py).

'dwt' is PINNED (round 5) against PyWavelets 1.1.1 -- the real package, found in this image's conda interpreter
(/opt/conda/bin/python3.9; oracle/make_golden_thirdparty.py -> tests/golden/thirdparty_pins.npz: forward, inverse, fp32 /
fp64, three sizes, sub-band impulses).  Haar level-3 `wavedec2` + `coeffs_to_array` (condition/utils.py:116-132):
dec_lo = [1/sqrt2, 1/sqrt2], dec_hi = [-1/sqrt2, 1/sqrt2] => cA[k] = (x[2k]+x[2k+1])/sqrt2, cD[k] = (x[2k]-x[2k+1])/sqrt2;
2-D keys are '<axis -2><axis -1>' with a=approx, d=detail; `coeffs_to_array` places a block by its key ('a' = leading half of
that axis, 'd' = trailing half): cA_L top-left and per level 'ad' (pywt's cV: approx down the rows, detail along the columns)
TOP-RIGHT, 'da' (cH) BOTTOM-LEFT, 'dd' (cD) bottom-right.  Rounds 1-4 restated the placement from the documentation's cH / cV
figure without the package and had 'ad' and 'da' exchanged; the pin found it.

'dct' follows scipy.fft.dctn(norm='ortho') over *all* axes of the batch-1
tensor (condition/utils.py:91-103); the length-1 batch axis is an identity, so
the batched form transforms axes (C,H,W) per sample.  scipy is importable, so
this one is pinned by golden vectors.
"""
import math
import numpy as np
import torch

_S = 1.0 / math.sqrt(2.0)


def _haar_step(x):
    """One 2-D Haar analysis step on the last two axes -> (aa, da, ad, dd)."""
    # axis -2 (H): pairs of rows
    a_h = (x[..., 0::2, :] + x[..., 1::2, :]) * _S
    d_h = (x[..., 0::2, :] - x[..., 1::2, :]) * _S
    # axis -1 (W): pairs of columns
    aa = (a_h[..., 0::2] + a_h[..., 1::2]) * _S
    ad = (a_h[..., 0::2] - a_h[..., 1::2]) * _S   # approx along H, detail along W
    da = (d_h[..., 0::2] + d_h[..., 1::2]) * _S   # detail along H, approx along W
    dd = (d_h[..., 0::2] - d_h[..., 1::2]) * _S
    return aa, da, ad, dd


def _haar_istep(aa, da, ad, dd):
    a_h = torch.empty(aa.shape[:-1] + (aa.shape[-1] * 2,), dtype=aa.dtype)
    d_h = torch.empty_like(a_h)
    a_h[..., 0::2] = (aa + ad) * _S
    a_h[..., 1::2] = (aa - ad) * _S
    d_h[..., 0::2] = (da + dd) * _S
    d_h[..., 1::2] = (da - dd) * _S
    x = torch.empty(a_h.shape[:-2] + (a_h.shape[-2] * 2, a_h.shape[-1]), dtype=aa.dtype)
    x[..., 0::2, :] = (a_h + d_h) * _S
    x[..., 1::2, :] = (a_h - d_h) * _S
    return x


def dwt_haar(x, level=3):
    """wavedec2(haar, level) + coeffs_to_array, Mallat layout, same shape as x."""
    out = torch.empty_like(x)
    cur = x
    n_h, n_w = x.shape[-2], x.shape[-1]
    for _ in range(level):
        aa, da, ad, dd = _haar_step(cur)
        h, w = aa.shape[-2], aa.shape[-1]
        out[..., :h, w:2 * w] = ad          # top-right: key 'ad'
        out[..., h:2 * h, :w] = da          # bottom-left: key 'da'
        out[..., h:2 * h, w:2 * w] = dd
        cur = aa
    out[..., :cur.shape[-2], :cur.shape[-1]] = cur
    return out


def idwt_haar(c, level=3):
    h, w = c.shape[-2] >> level, c.shape[-1] >> level
    cur = c[..., :h, :w]
    for _ in range(level):
        ad = c[..., :h, w:2 * w]
        da = c[..., h:2 * h, :w]
        dd = c[..., h:2 * h, w:2 * w]
        cur = _haar_istep(cur, da, ad, dd)
        h, w = h * 2, w * 2
    return cur


def dct_ortho(x):
    from scipy.fft import dctn
    return torch.Tensor(dctn(x.detach().numpy(), norm="ortho", axes=(1, 2, 3)))


def idct_ortho(x):
    from scipy.fft import idctn
    return torch.Tensor(idctn(x.detach().numpy(), norm="ortho", axes=(1, 2, 3)))


class OrthoTransform:
    """condition/utils.py:50-67."""

    def __init__(self, ortho_tf_type=None):
        self.ortho_tf_type = ortho_tf_type
        if ortho_tf_type not in (None, "dwt", "dct"):
            raise KeyError(ortho_tf_type)

    def __call__(self, x):
        if self.ortho_tf_type is None:
            return x
        return dwt_haar(x) if self.ortho_tf_type == "dwt" else dct_ortho(x)

    def inv(self, x):
        if self.ortho_tf_type is None:
            return x
        return idwt_haar(x) if self.ortho_tf_type == "dwt" else idct_ortho(x)
