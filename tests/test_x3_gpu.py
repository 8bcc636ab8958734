"""GPU: the split-precision conv arithmetic (KDIP_BF16X3: fp32 storage; per product one v_mfma_f32_32x32x16_bf16 on the bf16 heads of
both operands + two v_mfma_f32_32x32x16_f16 on the fp16-encoded tails, fp32 accumulation; csrc/conv.hip, Mma<f32x3_t>) against an fp64
reference of the same conv, next to the exact-f32 and plain bf16 kernels on the same data: the error ladder the mode is built on
(reference arithmetic: fp32 end to end, condition/diffpir_utils/utils_model.py:364 use_fp16=False, guided_diffusion/unet.py:182-213)."""
import ctypes as C
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _conv(L, dtype, x, w, b, ntaps, transpose_flip=0, storage_out=1):
    B, Cin, H, W = x.shape
    Cout = w.shape[0]
    n_out = w.shape[1] if transpose_flip else Cout
    n_in = Cout if transpose_flip else Cin
    y = torch.empty(B, n_out, H, W, device="cuda")
    xd = x.cuda()
    assert xd.shape[1] == n_in
    L.check(L.load().kdip_test_conv(L.stream(), dtype, ntaps, L.ptr(xd), B, w.shape[1], H, W, C.c_void_p(w.contiguous().data_ptr()),
                                    C.c_void_p(b.data_ptr()) if b is not None else None, Cout, transpose_flip, L.ptr(y), storage_out))
    return y.cpu()


@pytest.mark.parametrize("case", [(2, 128, 128, 64, 64, 9), (1, 256, 128, 32, 32, 9), (4, 512, 512, 8, 8, 9), (2, 256, 768, 16, 16, 1), (1, 128, 6, 64, 64, 9)])
def test_x3_error_ladder_vs_fp64(case):
    import kdip_amd._lib as L
    L.require_gpu()
    B, Cin, Cout, H, W, ntaps = case
    k = 3 if ntaps == 9 else 1
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(B, Cin, 1, 1, generator=g))      # per-channel scales over ~2 decades
    w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * ntaps) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=k // 2)
    scale = float(ref.abs().max())
    err = {name: float((_conv(L, code, x, w, b, ntaps).double() - ref).abs().max()) / scale for name, code in (("f32", 0), ("bf16", 1), ("bf16x3", 2))}
    print(f"\nconv {case}: max|err|/max|ref| f32 {err['f32']:.2e}  bf16x3 {err['bf16x3']:.2e}  bf16 {err['bf16']:.2e}")
    assert err["f32"] < 2e-6
    assert err["bf16x3"] < 2.5e-6                    # the same size as the exact-f32 kernel's fp32 accumulation error
    assert err["bf16x3"] < err["bf16"] / 1000        # and >= 1000 x below the plain bf16 kernel on the same data


def test_x3_dgrad_and_fp32_head():
    """the dgrad packing (flipped + transposed hi / lo planes) and the fp32-output head epilogue of the split-precision kernel"""
    import kdip_amd._lib as L
    L.require_gpu()
    g = torch.Generator().manual_seed(6)
    B, Cin, Cout, H, W = 2, 96, 160, 32, 32
    x = torch.randn(B, Cin, H, W, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    go = torch.randn(B, Cout, H, W, generator=g)
    ref = torch.autograd.grad((F.conv2d(x, w.double(), None, padding=1) * go.double()).sum(), x)[0]
    for so in (0, 1):
        y = _conv(L, 2, go, w, None, 9, transpose_flip=1, storage_out=so)
        e = float((y.double() - ref).abs().max() / ref.abs().max())
        assert e < 2.5e-6, (so, e)


@pytest.mark.parametrize("scale", [1e-7, 1e-3, 1.0, 1e4])
def test_x3_gradient_scale_window(scale):
    """Gradients have no natural scale: the dgrad launches centre the fp16 window of the A operand on max |cotangent| (amax_bits +
    ConvStats::x3_amax, as UNet::vjp_impl does), so the error relative to the output does not depend on the cotangent's magnitude."""
    import kdip_amd._lib as L
    L.require_gpu()
    g = torch.Generator().manual_seed(8)
    B, Cin, Cout, H, W = 1, 128, 128, 32, 32
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    go = torch.randn(B, Cout, H, W, generator=g) * scale
    x = torch.zeros(B, Cin, H, W, dtype=torch.float64, requires_grad=True)
    ref = torch.autograd.grad((F.conv2d(x, w.double(), None, padding=1) * go.double()).sum(), x)[0]
    y = _conv(L, 2, go, w, None, 9, transpose_flip=1)
    e = float((y.double() - ref).abs().max() / ref.abs().max())
    print(f"\nx3 dgrad, cotangent scale {scale:g}: max|err|/max|ref| {e:.2e}")
    assert e < 2.5e-6, (scale, e)


def test_x3_out_of_window_degrades_gracefully():
    """Forward inputs are taken at O(1) scale (no amax): elements far outside the fp16 window lose their cross terms and fall back
    towards the bf16 head (2^-9), they never produce inf / NaN (saturating conversions)."""
    import kdip_amd._lib as L
    L.require_gpu()
    g = torch.Generator().manual_seed(9)
    B, Cin, Cout, H, W = 1, 64, 64, 16, 16
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    for scale, bound in ((1e6, 5e-3), (1e-9, 5e-3), (30.0, 2.5e-6)):
        x = torch.randn(B, Cin, H, W, generator=g) * scale
        ref = F.conv2d(x.double(), w.double(), None, padding=1)
        y = _conv(L, 2, x, w, None, 9)
        assert torch.isfinite(y).all()
        e = float((y.double() - ref).abs().max() / ref.abs().max())
        print(f"\nx3 forward, input scale {scale:g}: max|err|/max|ref| {e:.2e}")
        assert e < bound, (scale, e)


def test_x3_unet_with_wide_dynamic_range():
    """The window assumptions of the split-precision mode on a network whose tensors are far from O(1): the tiny UNet with its second
    ResBlock convs scaled x 100 (the raw residual stream that the 1x1 skip convs and the next GroupNorms read reaches the hundreds; the
    backward gains of those convs spread the internal gradients of one VJP over many decades) and VJP cotangents of scale 1e-6 / 1 / 1e+3.
    Forward: bf16x3 = exact-f32 to the f32 mode's own accuracy.  VJP, default window (one power-of-two scale per VJP from max |cotangent|):
    the tails of the far-off layers fade -- the documented graceful degradation (5e-5 relative; plain bf16: 4e-2), finite, independent of
    the cotangent's own scale.  VJP with `set_x3_window("launch")` (every dgrad launch scales by a sampled max of its own input): back at
    1e-5 on this extreme network."""
    import kdip_amd.unet as ku
    from oracle import unet as ounet
    cfg = ounet.UNetConfig(**ounet.TINY)
    sd = {k: (v * 100.0 if ".out_layers.3.weight" in k else v) for k, v in ounet.init_state_dict(cfg, seed=0).items()}
    kw = dict(image_size=64, model_channels=32, num_res_blocks=1, attention_resolutions="32", channel_mult=(1, 2))
    ms = {}
    for dt in ("f32", "bf16x3", "bf16"):
        ms[dt] = ku.UNetModel(dtype=dt, **kw)
        ms[dt].load_state_dict(sd)
    g = torch.Generator().manual_seed(13)
    x = torch.randn(2, 3, 64, 64, generator=g).cuda()
    t = torch.tensor([100.0, 700.0]).cuda()
    outs = {dt: m.forward_raw(x, t, want_feature=True) for dt, m in ms.items()}
    feat = outs["f32"][2]
    assert float(feat.abs().max()) > 100.0, float(feat.abs().max())       # the stream really is far from O(1)
    e = float((outs["bf16x3"][0] - outs["f32"][0]).abs().max() / outs["f32"][0].abs().max())
    print(f"\nwide-range UNet: max |feature| {float(feat.abs().max()):.0f}; forward rel-max bf16x3 vs f32 {e:.2e}")
    assert e < 2e-5, e
    for scale in (1e-6, 1.0, 1e3):
        cot = (torch.randn(2, 6, 64, 64, generator=g) * scale).cuda()
        v = {dt: m.vjp(cot) for dt, m in ms.items()}
        v2 = ms["f32"].vjp(cot)
        ev = float((v["bf16x3"] - v["f32"]).abs().max() / v["f32"].abs().max())
        eb = float((v["bf16"] - v["f32"]).abs().max() / v["f32"].abs().max())
        en = float((v2 - v["f32"]).abs().max() / v["f32"].abs().max())
        print(f"wide-range UNet: VJP rel-max vs f32 at cotangent scale {scale:g}: bf16x3 {ev:.2e}, bf16 {eb:.2e}, f32 re-run {en:.2e}")
        assert torch.isfinite(v["bf16x3"]).all() and ev < 5e-4 and ev < eb / 100, (scale, ev, eb)
        ms["bf16x3"].set_x3_window("launch")
        try:
            ms["bf16x3"].forward_raw(x, t)
            el = float((ms["bf16x3"].vjp(cot) - v["f32"]).abs().max() / v["f32"].abs().max())
        finally:
            ms["bf16x3"].set_x3_window("vjp")
            ms["bf16x3"].forward_raw(x, t)
        print(f"                 ... per-launch window: bf16x3 {el:.2e}")
        assert el < 3e-5 and el < ev, (scale, el, ev)


def test_x3_saturation_flag():
    """The fp16 window of the split-precision convs is watched: `x3_saturated()` stays 0 while every staged operand (after its power-of-two
    scaling) is inside +-65504 -- i.e. every product carried its full precision -- and bit 0 is set by the launch that stages one outside
    (here: a forward whose input is scaled by 1e7, so the first conv's activations leave the window); bit 1 reports weights outside the
    window at pack time (|w| > 255.9)."""
    import kdip_amd.unet as ku
    from oracle import unet as ounet
    cfg = ounet.UNetConfig(**ounet.TINY)
    sd = ounet.init_state_dict(cfg, seed=0)
    kw = dict(image_size=64, model_channels=32, num_res_blocks=1, attention_resolutions="32", channel_mult=(1, 2))
    m = ku.UNetModel(dtype="bf16x3", **kw)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 64, 64, generator=g).cuda()
    t = torch.tensor([100.0, 700.0]).cuda()
    m.forward_raw(x, t)
    m.vjp(torch.randn(2, 6, 64, 64, generator=g).cuda() * 1e-4)
    assert m.x3_saturated() == 0
    m.forward_raw(x * 1e7, t)
    assert m.x3_saturated() & 1
    assert m.x3_saturated() == 0                     # (the query reset it)
    big = {k: (v * 1e4 if k == "input_blocks.0.0.weight" else v) for k, v in sd.items()}
    m2 = ku.UNetModel(dtype="bf16x3", **kw)
    m2.load_state_dict(big)
    assert m2.x3_saturated() & 2


@pytest.mark.parametrize("env", [{"KDIP_X3_TF2": "1"}, {"KDIP_X3_TF2": "1", "KDIP_STORE_DZ": "1"}, {"KDIP_GNB_FOLD": "0", "KDIP_STORE_DZ": "2"}])
def test_x3_optin_backward_fusions_match_reference(env):
    """The opt-in backward fusions of the fp32-storage modes (csrc/unet.hip: KDIP_X3_TF2 = GroupNorm-backward staging inside the split-precision
    dgrad conv -- measured slower, off by default; KDIP_STORE_DZ = dz left behind by the backward-statistics epilogue; KDIP_GNB_FOLD = 0 = the
    round-5 path without the GroupNorm-backward epilogue of the skip dgrad conv) are read once per process: run the mid-size REFERENCE capture
    (forward + autograd VJP + guided calls written by the imported reference, tests/test_mid_gpu.py) in a child interpreter with them set."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_mid_gpu.py"), "-x", "-q", "-m", "gpu"],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, (env, r.stdout[-2000:], r.stderr[-1000:])


def test_f16x3_conv_error_ladder():
    """The opt-in fp16-headed split (dtype "f16x3": three fp16 MFMAs per product, 11 + 11-bit operands; csrc/conv.hip Mma<f32h3_t>) against an
    fp64 conv on data inside its window: at least the accuracy of the bf16-headed split (measured 4.7e-7 vs 7e-7), forward and input-gradient
    (the dgrad operand is scaled by max |cotangent| as in the UNet's VJP), 3x3 and 1x1."""
    import kdip_amd._lib as L
    L.require_gpu()
    g = torch.Generator().manual_seed(5)
    for (B, Cin, Cout, H, W, nt) in ((2, 128, 128, 32, 32, 9), (1, 256, 64, 16, 16, 9), (2, 192, 128, 16, 16, 1)):
        k = 3 if nt == 9 else 1
        w = torch.randn(Cout, Cin, k, k, generator=g) / (Cin * nt) ** 0.5
        x = torch.randn(B, Cin, H, W, generator=g)
        ref = F.conv2d(x.double(), w.double(), None, padding=k // 2)
        e3 = float((_conv(L, 3, x, w, None, nt).double() - ref).abs().max() / ref.abs().max())
        e2 = float((_conv(L, 2, x, w, None, nt).double() - ref).abs().max() / ref.abs().max())
        print(f"\nf16x3 conv {Cin}->{Cout} @{H} taps {nt}: max|err|/max|ref| f16x3 {e3:.2e}  (bf16x3 {e2:.2e})")
        assert e3 < 1.2e-6, (Cin, Cout, nt, e3)
    for scale in (1e-3, 1.0, 1e4):                    # gradient operand: no natural scale, window follows max |g|
        gch = torch.randn(2, 128, 32, 32, generator=g) * scale
        w = torch.randn(128, 96, 3, 3, generator=g) / (96 * 9) ** 0.5
        ref = F.conv_transpose2d(gch.double(), w.double(), None, padding=1)
        y = _conv(L, 3, gch, w, None, 9, transpose_flip=1)
        e = float((y.double() - ref).abs().max() / ref.abs().max())
        print(f"f16x3 dgrad, cotangent scale {scale:g}: max|err|/max|ref| {e:.2e}")
        assert e < 1.2e-6, (scale, e)


@pytest.mark.parametrize("case", [(1, 128, 128, 256, 256, 0), (2, 64, 256, 128, 128, 0), (4, 128, 128, 64, 256, 1), (2, 96, 128, 128, 256, 1)])
def test_f16x3_row_reuse_tile(case):
    """The chip-filling fp16-headed 3x3 launches (>= 512 tiles of 128 px x 128 co) run the 1 x 4-wave tile on 32 x 4-pixel patches with the A
    fragments of a halo row shared by the three row taps (KDIP_H3_ROWREUSE, csrc/conv.hip): forward (bias) and input-gradient, square and
    wide maps, a ragged channel count -- against an fp64 conv, at the error of the other fp16-headed tiles, and finite / exact on the borders
    (an all-ones input with all-ones weights counts the taps inside the image)."""
    import kdip_amd._lib as L
    L.require_gpu()
    B, Cin, Cout, H, W, dgrad = case
    g = torch.Generator().manual_seed(17)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    if dgrad:
        go = torch.randn(B, Cout, H, W, generator=g) * 0.03
        ref = F.conv_transpose2d(go.double(), w.double(), None, padding=1)
        y = _conv(L, 3, go, w, None, 9, transpose_flip=1)
    else:
        x = torch.randn(B, Cin, H, W, generator=g)
        b = torch.randn(Cout, generator=g)
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
        y = _conv(L, 3, x, w, b, 9)
    e = float((y.double() - ref).abs().max() / ref.abs().max())
    print(f"\nf16x3 row-reuse tile {case}: max|err|/max|ref| {e:.2e}")
    assert e < 1.2e-6, (case, e)
    if not dgrad:
        ones = _conv(L, 3, torch.ones(1, Cin, H, W), torch.ones(Cout, Cin, 3, 3) / 64.0, None, 9)
        cnt = F.conv2d(torch.ones(1, 1, H, W), torch.ones(1, 1, 3, 3), padding=1) * (Cin / 64.0)
        assert torch.equal(ones, cnt.expand(1, Cout, H, W)), "tap count on the borders"


def test_f16x3_out_of_window_call_is_redone_bf16_headed():
    """dtype "f16x3" is exact-grade only inside the fp16 window of its head planes.  An operand beyond +-65504 (after the power-of-two scaling)
    raises bit 0 of kdip_unet_x3_saturated; UNetModel.guarded() polls it after every forward / VJP / fused guided call and redoes the
    flagged call on the bf16-headed weights the handle carries: the result is BITWISE the one a "bf16x3" handle produces, and the next
    in-window call runs fp16-headed again."""
    import kdip_amd.unet as ku
    from oracle import unet as ounet
    cfg = ounet.UNetConfig(**ounet.TINY)
    sd = ounet.init_state_dict(cfg, seed=0)
    kw = dict(image_size=64, model_channels=32, num_res_blocks=1, attention_resolutions="32", channel_mult=(1, 2))
    mb = ku.UNetModel(dtype="bf16x3", **kw).load_state_dict(sd)
    mh = ku.UNetModel(dtype="f16x3", **kw).load_state_dict(sd)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 64, 64, generator=g).cuda()
    t = torch.tensor([100.0, 700.0]).cuda()
    o1 = mh.forward_raw(x, t)[0].clone()
    assert mh.x3_fallbacks == 0 and not torch.equal(o1, mb.forward_raw(x, t)[0])      # a different arithmetic, inside its window
    ob = mb.forward_raw(x * 1e7, t)[0].clone()
    oh = mh.forward_raw(x * 1e7, t)[0].clone()
    assert mh.x3_fallbacks == 1 and torch.isfinite(oh).all() and torch.equal(oh, ob)
    # the LOW side of the window: with activations of 1e7 the GroupNorm backward scales every internal gradient by ~1e-7 of the cotangent
    # the VJP's window follows -- whole gradient tensors under the fp16 head's normal range (2^-14), where it would lose them.  Each launch
    # reports its largest staged operand, the end-of-pass check raises bit 2, and the VJP is redone bf16-headed as well.
    cot = torch.randn(2, 6, 64, 64, generator=g).cuda()
    vb = mb.vjp(cot).clone()
    vh = mh.vjp(cot).clone()
    assert mh.x3_fallbacks == 2 and torch.equal(vh, vb), float((vh - vb).abs().max() / vb.abs().max())
    o2 = mh.forward_raw(x, t)[0]
    mh.vjp(cot)
    assert mh.x3_fallbacks == 2 and torch.equal(o1, o2)
    mh.x3_guard = False                               # the caller polls itself (what graphs.py does around a replay)
    mh.forward_raw(x * 1e7, t)
    assert mh.x3_saturated() & 1 and mh.x3_fallbacks == 2
    mh.vjp(cot)
    assert mh.x3_saturated() & 4
    # policy: gradient tensors under the window are a property of the network -- with x3_auto_window (opt-in) the first flagged VJP makes every
    # later dgrad launch take its own scale (set_x3_window("launch") at the next forward) instead of a bf16-headed redo per call
    mh.x3_guard, mh.x3_auto_window = True, True
    mh.forward_raw(x * 1e7, t)
    mh.vjp(cot)
    assert mh._x3_window_pending and mh._x3_window == "vjp"
    n = mh.x3_fallbacks
    mh.forward_raw(x * 1e7, t)                        # (its activations still leave the window: redone bf16-headed; the switch happened before it)
    assert mh._x3_window == "launch" and mh.x3_window_switches == 1
    n = mh.x3_fallbacks
    vl = mh.vjp(cot)
    assert mh.x3_fallbacks == n and torch.isfinite(vl).all()      # per-launch scales: the fp16-headed VJP is inside its window now
    mf = ku.UNetModel(dtype="f32", **kw).load_state_dict(sd)
    mf.forward_raw(x * 1e7, t)
    vf = mf.vjp(cot)
    el, eb = float((vl - vf).abs().max() / vf.abs().max()), float((vb - vf).abs().max() / vf.abs().max())
    print(f"\nactivations of 1e7: VJP rel-max vs f32: f16x3 with per-launch windows {el:.2e}, bf16x3 (tails lost, head only) {eb:.2e}")
    assert el < max(eb, 1e-4)                         # at least what the bf16-headed arithmetic delivers out there
