"""GPU: the second-generation bf16 3x3 conv (csrc/conv3.hip) through the C-ABI test hook against a plain PyTorch fp32
reference of the same op on the same bf16-rounded operands: plain conv, fused GroupNorm+SiLU input staging (tf 1), fused
GroupNorm-backward input staging (tf 2), residual, fused nearest-x2 upsample of input / residual, and the GroupNorm forward /
backward statistics of the output accumulated in the epilogue.  Tolerances are written next to each check: the output is
stored as bf16 (relative rounding 2^-9), sums are compared relative to their scale."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def silu_grad(z):
    s = torch.sigmoid(z)
    return s * (1 + z * (1 - s))


def run_conv3(x, w, bias, Cout, x2=None, tf=0, tf_coef=None, res=None, in_ups=0, res_ups=0, st_mode=0, stx=None, st_coef=None,
              st_mr=None, transpose_flip=0, reps=1):
    import kdip_amd._lib as L
    lib = L.load()
    B = x.shape[0]
    H, W = (x.shape[2] * 2, x.shape[3] * 2) if in_ups else (x.shape[2], x.shape[3])
    Cin = w.shape[1]
    Co = w.shape[1] if transpose_flip else w.shape[0]
    y = torch.empty(B, Co, H, W, device="cuda")
    sums = torch.zeros(B, 32, 2, device="cuda", dtype=torch.float64) if st_mode else None
    us = C.c_float(0)
    wc = w.contiguous().cpu()
    bc = bias.contiguous().cpu() if bias is not None else None
    keep = [t.cuda().contiguous() if t is not None else None for t in (x, x2, tf_coef, res, stx, st_coef, st_mr)]
    L.check(lib.kdip_test_conv3(L.stream(), L.ptr(keep[0]), L.ptr(keep[1]), B, Cin, H, W, C.c_void_p(wc.data_ptr()),
                                C.c_void_p(bc.data_ptr()) if bc is not None else None, w.shape[0], transpose_flip, tf, L.ptr(keep[2]),
                                L.ptr(keep[3]), in_ups, res_ups, st_mode, L.ptr(keep[4]), L.ptr(keep[5]), L.ptr(keep[6]), L.ptr(y),
                                L.ptr(sums), reps, C.byref(us)))
    torch.cuda.synchronize()
    return y.cpu(), (sums.cpu() if sums is not None else None), us.value


def rel_err(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 128, 128, 32, 64), (1, 64, 256, 16, 32), (1, 32, 128, 8, 32)])
def test_conv3_plain_bias_residual(B, Cin, Cout, H, W):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    r = torch.randn(B, Cout, H, W, generator=g)
    y, _, _ = run_conv3(x, w, b, Cout, res=r)
    ref = F.conv2d(bf(x), bf(w), b, padding=1) + bf(r)
    assert rel_err(y, ref) < 1e-2          # bf16 output rounding (2^-9 relative) + fp32 accumulation-order noise


def test_conv3_dgrad_weights():
    """transpose_flip packing = the input-gradient of the conv (the VJP's dgrad convs use this kernel with wb)."""
    g = torch.Generator().manual_seed(1)
    B, Cin, Cout, H, W = 1, 128, 256, 16, 32          # forward conv Cin -> Cout; dgrad maps Cout -> Cin channels
    gy = torch.randn(B, Cout, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    y, _, _ = run_conv3(gy, w, None, Cout, transpose_flip=1)
    xr = torch.zeros(B, Cin, H, W, requires_grad=True)
    out = F.conv2d(xr, bf(w), None, padding=1)
    ref = torch.autograd.grad((out * bf(gy)).sum(), xr)[0]
    assert rel_err(y, ref) < 1e-2


def test_conv3_fused_groupnorm_silu_staging_and_forward_stats():
    """tf 1: A = silu(a*x + b) applied while staging (zero padding applies to the activated tensor), st_mode 1: (sum y, sum y^2)
    per (image, group) of the stored output."""
    g = torch.Generator().manual_seed(2)
    B, Cin, Cout, H, W = 2, 128, 128, 32, 32
    x = torch.randn(B, Cin, H, W, generator=g)
    coef = torch.stack([torch.rand(B, Cin, generator=g) + 0.5, torch.randn(B, Cin, generator=g) * 0.3], dim=-1)   # (a, b)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    y, sums, _ = run_conv3(x, w, b, Cout, tf=1, tf_coef=coef, st_mode=1)
    a_, b_ = coef[..., 0][:, :, None, None], coef[..., 1][:, :, None, None]
    A = bf(F.silu(a_ * bf(x) + b_))
    ref = F.conv2d(A, bf(w), b, padding=1)
    assert rel_err(y, ref) < 1.5e-2
    yr = y.double().view(B, 32, -1)                   # the kernel sums its fp32 values before the bf16 rounding of the store:
    s_ref = torch.stack([yr.sum(-1), (yr * yr).sum(-1)], dim=-1)      # vs the stored values the sums differ by rounding noise only
    assert float((sums - s_ref).abs().max() / s_ref.abs().max()) < 1e-3
    yf = ref.double().view(B, 32, -1)                 # and agree with the fp32 reference conv more closely
    s_f = torch.stack([yf.sum(-1), (yf * yf).sum(-1)], dim=-1)
    assert float((sums - s_f).abs().max() / s_f.abs().max()) < 2e-4


def test_conv3_fused_upsample_reads():
    g = torch.Generator().manual_seed(3)
    B, Cin, Cout, H, W = 1, 64, 128, 32, 64
    xh = torch.randn(B, Cin, H // 2, W // 2, generator=g)
    rh = torch.randn(B, Cout, H // 2, W // 2, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    y, _, _ = run_conv3(xh, w, None, Cout, res=rh, in_ups=1, res_ups=1)
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    ref = F.conv2d(up(bf(xh)), bf(w), None, padding=1) + up(bf(rh))
    assert rel_err(y, ref) < 1e-2


def test_conv3_groupnorm_backward_staging_and_backward_stats():
    """tf 2: A = a*dz - (k0 + k1*x2) applied while staging (dz = the staged tensor, as a st_mode-2 epilogue stores it); st_mode 2:
    the conv output dy is turned into dz = dy * silu'(a*stx + b) in the epilogue, dz is what is STORED, and the sums
    (sum a*dz, sum a*dz*xhat) w.r.t. the GroupNorm whose input is stx are accumulated."""
    g = torch.Generator().manual_seed(4)
    B, Cin, Cout, H, W = 2, 128, 128, 16, 32
    dzin = torch.randn(B, Cin, H, W, generator=g)
    x2 = torch.randn(B, Cin, H, W, generator=g)
    tfc = torch.stack([torch.rand(B, Cin, generator=g) + 0.5, torch.randn(B, Cin, generator=g) * 0.3,
                       torch.randn(B, Cin, generator=g) * 0.1, torch.randn(B, Cin, generator=g) * 0.1], dim=-1)   # (a, b, k0, k1)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    stx = torch.randn(B, Cout, H, W, generator=g)
    stc = torch.stack([torch.rand(B, Cout, generator=g) + 0.5, torch.randn(B, Cout, generator=g) * 0.3], dim=-1)
    mr = torch.stack([torch.randn(B, 32, generator=g) * 0.2, torch.rand(B, 32, generator=g) + 0.5], dim=-1)       # (mean, rstd)
    y, sums, _ = run_conv3(dzin, w, None, Cout, x2=x2, tf=2, tf_coef=tfc, st_mode=2, stx=stx, st_coef=stc, st_mr=mr)
    e = lambda t, i: t[..., i][:, :, None, None]
    A = bf(e(tfc, 0) * bf(dzin) - (e(tfc, 2) + e(tfc, 3) * bf(x2)))
    dy = F.conv2d(A, bf(w), None, padding=1)
    zz = e(stc, 0) * bf(stx) + e(stc, 1)
    ref = bf(dy) * silu_grad(zz)                      # the kernel rounds dy to bf16, applies silu', rounds the product
    assert rel_err(y, ref) < 2e-2
    # backward sums against the values the kernel stored (it sums the fp32 products before the final rounding)
    cpg = Cout // 32
    adz = (e(stc, 0) * y).double()
    mean = mr[..., 0].repeat_interleave(cpg, dim=1)[:, :, None, None].double()
    rstd = mr[..., 1].repeat_interleave(cpg, dim=1)[:, :, None, None].double()
    xhat = (bf(stx).double() - mean) * rstd
    t1 = adz.view(B, 32, -1).sum(-1)
    t2 = (adz * xhat).view(B, 32, -1).sum(-1)
    s_ref = torch.stack([t1, t2], dim=-1)
    assert float((sums - s_ref).abs().max() / s_ref.abs().max()) < 2e-3      # fp32 partial sums + rounding of the stored dz
