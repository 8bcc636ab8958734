"""GPU: the third-generation bf16 3x3 conv (csrc/conv4.hip: one 8-wave block per CU, 16 x 32-pixel x 128-channel tiles, two wave
groups in ping-pong) through the same C-ABI test hook as conv3 with the kernel generation forced to 4, against a plain PyTorch
fp32 reference of the same op on the same bf16-rounded operands -- every fusion mode the UNet executor instantiates, shapes
with several tiles per block (persistent loop, staging pipeline across tile boundaries, image changes inside a block's tile
list), one and several 32-channel chunks, one and two 128-channel output blocks.  Tolerances as in test_conv3_gpu.py.
Also: conv4 == conv3 on identical inputs to bf16 rounding."""
import pytest
import torch
import torch.nn.functional as F

from test_conv3_gpu import run_conv3, bf, silu_grad, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def force_conv4():
    import kdip_amd._lib as L
    lib = L.load()
    L.check(lib.kdip_debug_conv_generation(4))
    yield
    L.check(lib.kdip_debug_conv_generation(0))


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 128, 128, 32, 64), (1, 64, 256, 16, 32), (1, 32, 128, 16, 32), (3, 96, 128, 64, 32),
                                             (40, 32, 128, 64, 64)])
def test_conv4_plain_bias_residual(B, Cin, Cout, H, W):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    r = torch.randn(B, Cout, H, W, generator=g)
    y, _, _ = run_conv3(x, w, b, Cout, res=r)
    ref = F.conv2d(bf(x), bf(w), b, padding=1) + bf(r)
    assert rel_err(y, ref) < 1e-2          # bf16 output rounding (2^-9 relative) + fp32 accumulation-order noise
    y2, _, _ = run_conv3(x, w, b, Cout)    # no residual, plain epilogue
    assert rel_err(y2, F.conv2d(bf(x), bf(w), b, padding=1)) < 1e-2


def test_conv4_many_tiles_per_block_equals_conv3():
    """640 tiles over 256 persistent blocks (2 - 3 tiles each, image changes inside the lists): conv4 == conv3 to bf16 rounding."""
    import kdip_amd._lib as L
    g = torch.Generator().manual_seed(7)
    B, Cin, Cout, H, W = 10, 64, 256, 64, 128
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    coef = torch.stack([torch.rand(B, Cin, generator=g) + 0.5, torch.randn(B, Cin, generator=g) * 0.3], dim=-1)
    y4, s4, _ = run_conv3(x, w, b, Cout, tf=1, tf_coef=coef, st_mode=1)
    L.check(L.load().kdip_debug_conv_generation(3))
    y3, s3, _ = run_conv3(x, w, b, Cout, tf=1, tf_coef=coef, st_mode=1)
    assert rel_err(y4, y3) < 8e-3          # same operands, same MFMA order per output; only fp32 epilogue / statistics order differs
    assert float((s4 - s3).abs().max() / s3.abs().max()) < 1e-5
    a_, b_ = coef[..., 0][:, :, None, None], coef[..., 1][:, :, None, None]
    ref = F.conv2d(bf(F.silu(a_ * bf(x) + b_)), bf(w), b, padding=1)
    assert rel_err(y4, ref) < 1.5e-2


def test_conv4_dgrad_weights():
    g = torch.Generator().manual_seed(1)
    B, Cin, Cout, H, W = 1, 128, 256, 32, 32          # forward conv Cin -> Cout; dgrad maps Cout -> Cin channels
    gy = torch.randn(B, Cout, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    y, _, _ = run_conv3(gy, w, None, Cout, transpose_flip=1)
    xr = torch.zeros(B, Cin, H, W, requires_grad=True)
    out = F.conv2d(xr, bf(w), None, padding=1)
    ref = torch.autograd.grad((out * bf(gy)).sum(), xr)[0]
    assert rel_err(y, ref) < 1e-2


def test_conv4_fused_groupnorm_silu_staging_and_forward_stats():
    g = torch.Generator().manual_seed(2)
    B, Cin, Cout, H, W = 2, 128, 128, 32, 32
    x = torch.randn(B, Cin, H, W, generator=g)
    coef = torch.stack([torch.rand(B, Cin, generator=g) + 0.5, torch.randn(B, Cin, generator=g) * 0.3], dim=-1)   # (a, b)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    b = torch.randn(Cout, generator=g)
    r = torch.randn(B, Cout, H, W, generator=g)
    for res in (None, r):
        y, sums, _ = run_conv3(x, w, b, Cout, tf=1, tf_coef=coef, st_mode=1, res=res)
        a_, b_ = coef[..., 0][:, :, None, None], coef[..., 1][:, :, None, None]
        A = bf(F.silu(a_ * bf(x) + b_))
        ref = F.conv2d(A, bf(w), b, padding=1) + (bf(res) if res is not None else 0)
        assert rel_err(y, ref) < 1.5e-2
        yf = ref.double().view(B, 32, -1)
        s_f = torch.stack([yf.sum(-1), (yf * yf).sum(-1)], dim=-1)
        assert float((sums - s_f).abs().max() / s_f.abs().max()) < 2e-4


def test_conv4_fused_upsample_reads():
    g = torch.Generator().manual_seed(3)
    B, Cin, Cout, H, W = 1, 64, 128, 32, 64
    xh = torch.randn(B, Cin, H // 2, W // 2, generator=g)
    rh = torch.randn(B, Cout, H // 2, W // 2, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    y, _, _ = run_conv3(xh, w, None, Cout, res=rh, in_ups=1, res_ups=1)
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    ref = F.conv2d(up(bf(xh)), bf(w), None, padding=1) + up(bf(rh))
    assert rel_err(y, ref) < 1e-2


@pytest.mark.parametrize("tf", [0, 2])
def test_conv4_groupnorm_backward_staging_and_backward_stats(tf):
    g = torch.Generator().manual_seed(4)
    B, Cin, Cout, H, W = 2, 128, 128, 32, 32
    dzin = torch.randn(B, Cin, H, W, generator=g)
    x2 = torch.randn(B, Cin, H, W, generator=g)
    tfc = torch.stack([torch.rand(B, Cin, generator=g) + 0.5, torch.randn(B, Cin, generator=g) * 0.3,
                       torch.randn(B, Cin, generator=g) * 0.1, torch.randn(B, Cin, generator=g) * 0.1], dim=-1)   # (a, b, k0, k1)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    stx = torch.randn(B, Cout, H, W, generator=g)
    stc = torch.stack([torch.rand(B, Cout, generator=g) + 0.5, torch.randn(B, Cout, generator=g) * 0.3], dim=-1)
    mr = torch.stack([torch.randn(B, 32, generator=g) * 0.2, torch.rand(B, 32, generator=g) + 0.5], dim=-1)       # (mean, rstd)
    kw = dict(x2=x2, tf=2, tf_coef=tfc) if tf == 2 else {}
    y, sums, _ = run_conv3(dzin, w, None, Cout, st_mode=2, stx=stx, st_coef=stc, st_mr=mr, **kw)
    e = lambda t, i: t[..., i][:, :, None, None]
    A = bf(e(tfc, 0) * bf(dzin) - (e(tfc, 2) + e(tfc, 3) * bf(x2))) if tf == 2 else bf(dzin)
    dy = F.conv2d(A, bf(w), None, padding=1)
    zz = e(stc, 0) * bf(stx) + e(stc, 1)
    ref = bf(dy) * silu_grad(zz)
    assert rel_err(y, ref) < 2e-2
    cpg = Cout // 32
    adz = (e(stc, 0) * y).double()
    mean = mr[..., 0].repeat_interleave(cpg, dim=1)[:, :, None, None].double()
    rstd = mr[..., 1].repeat_interleave(cpg, dim=1)[:, :, None, None].double()
    xhat = (bf(stx).double() - mean) * rstd
    s_ref = torch.stack([adz.view(B, 32, -1).sum(-1), (adz * xhat).view(B, 32, -1).sum(-1)], dim=-1)
    assert float((sums - s_ref).abs().max() / s_ref.abs().max()) < 4e-3      # fp32 partial sums + rounding of the stored dz (cancelling sums)
    import kdip_amd._lib as L                          # and the same sums as the second-generation kernel (same per-lane arithmetic)
    L.check(L.load().kdip_debug_conv_generation(3))
    y3, sums3, _ = run_conv3(dzin, w, None, Cout, st_mode=2, stx=stx, st_coef=stc, st_mr=mr, **kw)
    L.check(L.load().kdip_debug_conv_generation(4))
    assert rel_err(y, y3) < 8e-3 and float((sums - sums3).abs().max() / sums3.abs().max()) < 1e-5
    if tf == 2:                                        # the plain-epilogue dgrad with the GroupNorm-backward staging (tf 2, st 0)
        y0, _, _ = run_conv3(dzin, w, None, Cout, **kw)
        assert rel_err(y0, dy) < 1.5e-2
