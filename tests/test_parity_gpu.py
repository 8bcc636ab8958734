"""GPU: parity of the HIP path against (a) the committed golden vectors captured from the
reference and (b) the CPU oracle on the same seeded inputs, through the reference-shaped
Python surface (which calls the C ABI).

Tolerances (max-abs on images in [-1,1]):
  * f32 mode (exact-f32 MFMA, fp32 activations): 2e-3 on guided x0 estimates.  The guided call
    amplifies UNet round-off by sigma^2 / (sigma_s^2 + C) (x400 at sigma_s = 0.05), and the CG
    branch is only tol=1e-4 accurate in the reference itself.
  * bf16x3 mode (split-precision convs, csrc/conv.hip Mma<f32x3_t>): the same bounds as f32 everywhere.
  * bf16 mode (throughput): reported as PSNR between HIP and reference outputs; floors = the measured
    minima minus 5 dB per operator (bf16 operand rounding, 8 mantissa bits, through ~40 conv layers and
    the VJP; a 10 x numerical regression costs 20 dB).
"""
import numpy as np
import pytest
import torch

from helpers import op_cfgs, synthetic_recon_mse, psnr_db

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def tiny():
    import kdip_amd.unet as ku
    from oracle import unet as ounet
    cfg = ounet.UNetConfig(**ounet.TINY)
    sd = ounet.init_state_dict(cfg, seed=0, out_cov=True)
    models = {}
    for dt in ("f32", "bf16", "bf16x3", "f16x3"):
        m = ku.UNetModel(image_size=64, model_channels=32, num_res_blocks=1, attention_resolutions="32",
                         channel_mult=(1, 2), dtype=dt)
        m.load_state_dict(sd)
        models[dt] = m
    return models, ku.GaussianDiffusionTables(), sd, cfg


def make_ops(name, gold):
    """HIP operator + oracle operator with identical mask / measurement."""
    import kdip_amd.measurements as km
    from oracle import operators as oops
    g = gold("operators")
    np.random.seed(0)
    hop = km.get_operator(name, device="cuda", **op_cfgs(64)[name])
    np.random.seed(0)
    oop = oops.get_operator(name, **op_cfgs(64)[name])
    y, yf = T(g[f"{name}.y"]), T(g[f"{name}.y_flat"])
    torch.manual_seed(2)
    oop.forward(T(g["x0"]).clone(), flatten=True)       # refresh oracle pre_calculated for this y
    return hop, oop, (y, yf), T(g["x0"])


# ------------------------------------------------------------------------- operators ----
@pytest.mark.parametrize("name", ["gaussian_blur", "motion_blur", "super_resolution", "inpainting"])
def test_operator_golden(gold, name):
    g = gold("operators")
    hop, oop, (y, yf), x0 = make_ops(name, gold)
    xd = x0.cuda()
    ynl = hop.forward(xd, noiseless=True)
    assert float((ynl.cpu() - T(g[f"{name}.y_noiseless"])).abs().max()) < 2e-6
    aty = hop.transpose(yf.cuda(), flatten=True)
    assert float((aty.cpu() - T(g[f"{name}.ATy"])).abs().max()) < 2e-6
    y2, yf2 = hop.forward(xd, flatten=True)
    assert yf2.shape == yf.shape and y2.shape == y.shape
    if name == "inpainting":
        # bit-exact: mask, gather indices, and gathered values of a noiseless forward
        bits = np.unpackbits(g["inpainting.mask_bits"])[:64 * 64].reshape(64, 64)
        assert np.array_equal(hop.mask[0, 0].cpu().numpy().astype(np.uint8), bits)
        ynl2, flat = hop.forward(xd, flatten=True, noiseless=True)
        ref_flat = oop.forward(x0, flatten=True, noiseless=True)[1]
        assert torch.equal(flat.cpu(), ref_flat)
        back = hop.transpose(flat, flatten=True)
        assert torch.equal(back.cpu(), oop.transpose(ref_flat, flatten=True))
    else:
        FB = hop.pre_calculated[0]
        assert float((torch.view_as_real(FB.cpu()) - torch.view_as_real(T(g[f"{name}.FB"]))).abs().max()) < 2e-5


def test_mask_256_bit_exact(gold):
    import kdip_amd.measurements as km
    g = gold("operators")
    np.random.seed(0)
    op = km.get_operator("inpainting", device="cuda", sigma_s=0.05,
                         mask_opt=dict(mask_type="random", mask_prob_range=(0.5, 0.5), image_size=256))
    bits = np.unpackbits(g["inpainting.mask256_bits"]).reshape(256, 256)
    assert np.array_equal(op.mask[0, 0].cpu().numpy().astype(np.uint8), bits)
    idx = op._idx.cpu()
    c, rem = idx // 65536, idx % 65536
    first = torch.stack([c, rem // 256, rem % 256])[:, :64]
    assert np.array_equal(first.numpy(), g["inpainting.mask256_first_idx"])
    assert idx.numel() == 3 * 32768
    # gather -> scatter round trip at full size and batch 3 is the identity on kept pixels
    x = torch.randn(3, 3, 256, 256, device="cuda")
    y, flat = op.forward(x, flatten=True, noiseless=True)
    assert torch.equal(op.transpose(flat, flatten=True), y)


def test_sr_resizer_256(gold):
    import kdip_amd.measurements as km
    from helpers import smooth_image
    g = gold("operators")
    op = km.get_operator("super_resolution", device="cuda", in_shape=(1, 3, 256, 256), scale_factor=4, sigma_s=0.05)
    y = op.forward(smooth_image(1, 256, 1).cuda(), noiseless=True)
    assert float((y.cpu() - T(g["sr256.y_noiseless"])).abs().max()) < 2e-6
    # adjoint identity <A x, r> == <x, A^T r> for the Resizer pair
    x = torch.randn(2, 3, 256, 256, device="cuda"); r = torch.randn(2, 3, 64, 64, device="cuda")
    lhs = float((op.forward(x, noiseless=True) * r).sum()); rhs = float((x * op.forward_adjoint(r)).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))


@pytest.mark.parametrize("name", ["gaussian_blur", "motion_blur"])
def test_blur_adjoint_and_fullsize(name):
    """<A x, y> == <x, A^T y> at 256x256, batch 2; spatial LDS stencil == FFT model."""
    import kdip_amd.measurements as km
    from oracle import operators as oops
    cfg = dict(op_cfgs(256)[name])
    op = km.get_operator(name, device="cuda", **cfg)
    x = torch.randn(2, 3, 256, 256, device="cuda"); y = torch.randn(2, 3, 256, 256, device="cuda")
    ax, aty = op.forward(x, noiseless=True), op.transpose(y)
    lhs, rhs = float((ax * y).sum()), float((x * aty).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))
    oop = oops.get_operator(name, **cfg)
    assert float((ax.cpu() - oop.forward(x.cpu(), noiseless=True)).abs().max()) < 5e-6
    assert float((aty.cpu() - oop.transpose(y.cpu())).abs().max()) < 5e-6


def test_transforms(gold):
    from kdip_amd.transforms import OrthoTransform
    from oracle import transforms as otf
    g = gold("transforms")
    x = T(g["x"]).cuda()
    dct = OrthoTransform("dct")
    assert float((dct(x).cpu() - T(g["dct"])).abs().max()) < 2e-5
    assert float((dct.inv(T(g["dct"]).cuda()).cpu() - T(g["x"])).abs().max()) < 2e-5
    dwt = OrthoTransform("dwt")
    xb = torch.randn(2, 3, 256, 256)
    w = dwt(xb.cuda())
    assert float((w.cpu() - otf.dwt_haar(xb)).abs().max()) < 1e-6
    assert float((dwt.inv(w).cpu() - xb).abs().max()) < 2e-6
    assert OrthoTransform(None)(x) is x
    with pytest.raises(KeyError):
        OrthoTransform("bogus")


# -------------------------------------------------------------------------- solvers ----
@pytest.mark.parametrize("name", ["gaussian_blur", "motion_blur", "super_resolution", "inpainting"])
@pytest.mark.parametrize("ortho", [None, "dwt", "dct"])
def test_solver_vs_oracle(gold, name, ortho):
    """closed form and CG branch (per-pixel variance, optional basis) against the oracle."""
    import kdip_amd.condition as kc
    from kdip_amd.transforms import OrthoTransform
    from oracle import solvers as osol, transforms as otf
    hop, oop, (y, yf), x0 = make_ops(name, gold)
    g = torch.Generator().manual_seed(4)
    B = 2
    x0m = (x0 + 0.1 * torch.randn(B, 3, 64, 64, generator=g)).clamp(-1, 1)
    yb = y.expand(B, -1, -1, -1).contiguous()
    solver = kc.__MAT_SOLVER__[name]
    # scalar variance -> closed form
    v = torch.tensor([0.3])
    ref = osol.MAT_SOLVER[name](oop, yb, x0m, v, otf.OrthoTransform(ortho))
    out = solver(hop, yb.cuda(), x0m.cuda(), v, OrthoTransform(ortho))
    assert float((out.cpu() - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max()))
    # tensor variance -> CG (reference accuracy: tol 1e-4 on the residual)
    vt = (0.02 + 0.2 * torch.rand(B, 3, 64, 64, generator=g))
    stats = {}
    ref = osol.MAT_SOLVER[name](oop, yb, x0m, vt, otf.OrthoTransform(ortho), cg_stats=stats)
    out = solver(hop, yb.cuda(), x0m.cuda(), vt.cuda(), OrthoTransform(ortho))
    scale = max(1.0, float(ref.abs().max()))
    assert float((out.cpu() - ref).abs().max()) < 5e-3 * scale
    assert all(i == 0 for i in hop.cg_info)
    assert max(abs(a - int(b)) for a, b in zip(hop.cg_iters, stats["iters"])) <= 1


# --------------------------------------------------------------------- guided calls ----
GUIDED = [("I", "convert", {}), ("II", "convert", {}), ("II", "pgdm", {}), ("dps", "dps", dict(zeta=1.0)),
          ("pgdm", "pgdm", {}), ("I", "analytic", {}), ("diffpir", "diffpir", dict(lambda_=7.0)),
          ("uncond", "convert", {}), ("dps+mle", "convert", dict(zeta=1.0))]


BF16_CALL_PSNR_FLOOR = {"gaussian_blur": 39.0, "motion_blur": 29.5, "super_resolution": 33.0, "inpainting": 30.0}      # measured minima: 44.0 / 34.5 / 38.0 / 35.2 dB


@pytest.mark.parametrize("name", ["gaussian_blur", "motion_blur", "super_resolution", "inpainting"])
def test_guided_calls_golden(gold, tiny, name):
    import kdip_amd.condition as kc
    models, D, sd, cfg = tiny
    g = gold("guided_calls")
    hop, oop, (y, yf), x0 = make_ops(name, gold)
    meas = (y.cuda(), yf.cuda())
    worst = {"f32": 0.0, "bf16x3": 0.0, "f16x3": 0.0}
    min_psnr = 1e9
    for guidance, cov, extra in GUIDED:
        for sigma_v in (1.5, 0.12):
            x = (x0 + sigma_v * torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(11))).cuda()
            ref = T(g[f"{name}|{guidance}|{cov}|{sigma_v}"])
            for dt in ("f32", "bf16x3", "f16x3", "bf16"):
                m = kc.ConditionOpenAIDenoiser(inner_model=models[dt], diffusion=D, x0_cov_type=cov,
                                               recon_mse=synthetic_recon_mse(), operator=hop, measurement=meas,
                                               guidance=guidance, zeta=extra.get("zeta"), lambda_=extra.get("lambda_"),
                                               mle_sigma_thres=0.2, device="cuda").eval()
                hat = m(x, torch.tensor([sigma_v], device="cuda")).cpu()
                if dt != "bf16":       # exact-f32 MFMA and split-precision (bf16x3) convs: the same bound
                    err = float((hat - ref).abs().max())
                    worst[dt] = max(worst[dt], err)
                    assert err < 2e-3, (dt, name, guidance, cov, sigma_v, err)
                else:       # bf16: per-operator floor = the measured minimum over the 18 calls minus 5 dB (a 10 x numerical regression costs 20 dB)
                    p = psnr_db(hat, ref)
                    min_psnr = min(min_psnr, p)
                    assert p > BF16_CALL_PSNR_FLOOR[name], (name, guidance, cov, sigma_v, p)
    print(f"\n[{name}] f32 worst max-abs {worst['f32']:.2e}; bf16x3 worst max-abs {worst['bf16x3']:.2e}; f16x3 worst max-abs {worst['f16x3']:.2e} ({models['f16x3'].x3_fallbacks} calls redone bf16-headed so far); bf16 min PSNR vs reference {min_psnr:.1f} dB")


def test_guided_calls_v2_golden(gold, tiny):
    import kdip_amd.condition as kc
    import kdip_amd.external as ke
    models, D, sd, cfg = tiny
    g = gold("guided_calls_v2")
    for name in ("gaussian_blur", "inpainting", "super_resolution"):
        hop, oop, (y, yf), x0 = make_ops(name, gold)
        meas = (y.cuda(), yf.cuda())
        for guidance in ("I", "II"):
            for sigma_v in (1.5, 0.12):
                x = (x0 + sigma_v * torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(11))).cuda()
                ref = T(g[f"{name}|{guidance}|v2|{sigma_v}"])
                for dt in ("f32", "bf16x3", "f16x3"):       # (the out_cov 1x1 head and the fractional-t path in both exact modes)
                    den = ke.OpenAIDenoiserV2(models[dt], D)
                    m = kc.ConditionOpenAIDenoiserV2(den, operator=hop, measurement=meas, guidance=guidance,
                                                     mle_sigma_thres=1.0, device="cuda").eval()
                    hat = m(x, torch.tensor([sigma_v], device="cuda")).cpu()
                    assert float((hat - ref).abs().max()) < 2e-3, (dt, name, guidance, sigma_v)


def test_guided_calls_v2_transform_bases_golden(gold, tiny):
    """HIP path vs the REFERENCE's own captures of the V2 denoiser with a DWT / DCT covariance basis (BASELINE configs[4]'s machinery;
    fixtures written by the reference with the real PyWavelets 1.1.1: oracle/make_golden_v2dwt.py): Type-I and Type-II, scalar variance
    (sigma 1.5) and learned theta-variance with on-device CG in the transform basis (sigma 0.5, 0.12), f32 and bf16x3."""
    import kdip_amd.condition as kc
    import kdip_amd.external as ke
    models, D, sd, cfg = tiny
    g = gold("guided_calls_v2_ot")
    worst = {"f32": 0.0, "bf16x3": 0.0, "f16x3": 0.0}
    for basis in ("dwt", "dct"):
        for name in ("gaussian_blur", "inpainting", "super_resolution"):
            hop, oop, (y, yf), x0 = make_ops(name, gold)
            meas = (y.cuda(), yf.cuda())
            for guidance in ("I", "II"):
                for sigma_v in (1.5, 0.5, 0.12):
                    x = (x0 + sigma_v * torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(11))).cuda()
                    ref = T(g[f"{name}|{guidance}|v2|{basis}|{sigma_v}"])
                    for dt in ("f32", "bf16x3", "f16x3"):
                        den = ke.OpenAIDenoiserV2(models[dt], D, ortho_tf_type=basis)
                        m = kc.ConditionOpenAIDenoiserV2(den, operator=hop, measurement=meas, guidance=guidance, mle_sigma_thres=1.0,
                                                         device="cuda", ortho_tf_type=basis).eval()
                        hat = m(x, torch.tensor([sigma_v], device="cuda")).cpu()
                        e = float((hat - ref).abs().max())
                        worst[dt] = max(worst[dt], e)
                        assert e < 2e-3, (dt, basis, name, guidance, sigma_v, e)
    print(f"\nV2 + DWT / DCT bases vs reference captures: worst max-abs f32 {worst['f32']:.2e}, bf16x3 {worst['bf16x3']:.2e}, f16x3 {worst['f16x3']:.2e}")


def test_v2_dwt_autoI_vs_oracle(gold, tiny):
    """config-5 shape of the path: v2 denoiser, DWT basis, autoI (= Type-I gradient), low sigma -> CG
    with DWT in the matvec.  pywt's transform is pinned (tests/test_thirdparty_pins.py), GPyTorch's likelihood is not; checked against the oracle restatement."""
    import kdip_amd.condition as kc
    import kdip_amd.external as ke
    from oracle import condition as ocond
    models, D, sd, cfg = tiny
    hop, oop, (y, yf), x0 = make_ops("gaussian_blur", gold)
    B = 2
    yb = y.expand(B, -1, -1, -1).contiguous()
    for sigma_v in (1.5, 0.3):
        x = x0 + sigma_v * torch.randn(B, 3, 64, 64, generator=torch.Generator().manual_seed(12))
        om = ocond.GuidedDenoiser(sd, cfg, oop, (yb, yb.flatten(1)), "autoI", mle_sigma_thres=1.0, v2=True, ortho_tf_type="dwt")
        ref = om(x, torch.full((B,), sigma_v))
        den = ke.OpenAIDenoiserV2(models["f32"], D, ortho_tf_type="dwt")
        m = kc.ConditionOpenAIDenoiserV2(den, operator=hop, measurement=(yb.cuda(), yb.flatten(1).cuda()), guidance="autoI",
                                         mle_sigma_thres=1.0, device="cuda", ortho_tf_type="dwt").eval()
        hat = m(x.cuda(), torch.full((B,), sigma_v, device="cuda")).cpu()
        assert float((hat - ref).abs().max()) < 3e-3, sigma_v


@pytest.mark.parametrize("dt", ["f32", "bf16x3"])
def test_batch_semantics(gold, tiny, dt):
    """B independent problems: a batch of 3 equals three batch-1 calls (f32 and bf16x3 modes; in bf16x3 the batch also changes the
    per-VJP fp16 window -- max |cotangent| runs over the whole batch -- which must not matter at this accuracy)."""
    import kdip_amd.condition as kc
    models, D, sd, cfg = tiny
    hop, oop, (y, yf), x0 = make_ops("gaussian_blur", gold)
    g = torch.Generator().manual_seed(21)
    ys = (y + 0.05 * torch.randn(3, 3, 64, 64, generator=g)).cuda()
    xs = (x0 + 0.12 * torch.randn(3, 3, 64, 64, generator=g)).cuda()
    sig = torch.full((3,), 0.12, device="cuda")
    m = kc.ConditionOpenAIDenoiser(inner_model=models[dt], diffusion=D, x0_cov_type="convert", recon_mse=None,
                                   operator=hop, measurement=(ys, ys.flatten(1)), guidance="I", device="cuda")
    full = m(xs, sig)
    for b in range(3):
        mb = kc.ConditionOpenAIDenoiser(inner_model=models[dt], diffusion=D, x0_cov_type="convert", recon_mse=None,
                                        operator=hop, measurement=(ys[b:b + 1], ys[b:b + 1].flatten(1)), guidance="I", device="cuda")
        one = mb(xs[b:b + 1], sig[:1])
        assert float((one - full[b:b + 1]).abs().max()) < 2e-4


@pytest.mark.parametrize("name,guidance,cov,kw", [("gaussian_blur", "I", "convert", {}), ("inpainting", "dps", "dps", {"zeta": 1.0}),
                                                  ("super_resolution", "II", "pgdm", {})])
def test_shared_measurement_broadcast(gold, tiny, name, guidance, cov, kw):
    """A batch-1 measurement is shared by all B samples of a call (B posterior samples of one measurement, the
    harness' `-n`): identical to passing the measurement repeated B times; a mismatching batch is an error."""
    import kdip_amd.condition as kc
    models, D, sd, cfg = tiny
    hop, oop, (y, yf), x0 = make_ops(name, gold)
    y, yf = y.cuda(), yf.cuda()
    g = torch.Generator().manual_seed(5)
    xs = (x0 + 0.5 * torch.randn(3, 3, 64, 64, generator=g)).cuda()
    sig = torch.full((3,), 0.5, device="cuda")
    mk = lambda meas: kc.ConditionOpenAIDenoiser(inner_model=models["f32"], diffusion=D, x0_cov_type=cov, recon_mse=None, operator=hop,
                                                 measurement=meas, guidance=guidance, device="cuda", **kw)
    shared = mk((y, yf))(xs, sig)
    rep = mk((y.repeat(3, 1, 1, 1), yf.repeat(3, 1)))(xs, sig)
    assert float((shared - rep).abs().max()) < 1e-6
    with pytest.raises(ValueError):
        mk((y.repeat(2, 1, 1, 1), yf.repeat(2, 1)))(xs, sig)


def test_box_inpainting_vs_oracle(gold, tiny):
    """Box-mask inpainting (mask_type 'box', the centred box of measurements.py:300-320): operator, flatten / transpose paths
    and a Type-I + Convert guided call (CG branch) against the oracle."""
    import numpy as np
    import kdip_amd.condition as kc
    import kdip_amd.measurements as km
    from oracle import operators as oops, condition as ocond
    models, D, sd, cfg = tiny
    opt = dict(mask_type="box", mask_len_range=(20, 36), image_size=64)
    np.random.seed(3); hop = km.get_operator("inpainting", device="cuda", sigma_s=0.05, mask_opt=opt)
    np.random.seed(3); oop = oops.get_operator("inpainting", sigma_s=0.05, mask_opt=opt)
    assert torch.equal(hop.mask.cpu(), oop.mask)
    x0 = T(gold("operators")["x0"])
    torch.manual_seed(2); y_o, yf_o = oop.forward(x0.clone(), flatten=True)
    y_h = hop.forward(x0.cuda(), noiseless=True)
    assert float((y_h.cpu() - oop.forward(x0.clone(), noiseless=True)).abs().max()) < 1e-6
    at = hop.transpose(yf_o.cuda(), flatten=True)
    assert float((at.cpu() - oop.transpose(yf_o, flatten=True)).abs().max()) < 1e-6
    x = x0 + 0.12 * torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(11))
    sig = torch.tensor([0.12])
    ref = ocond.GuidedDenoiser(sd, cfg, oop, (y_o, yf_o), "I", x0_cov_type="convert")(x, sig)
    m = kc.ConditionOpenAIDenoiser(inner_model=models["f32"], diffusion=D, x0_cov_type="convert", recon_mse=None, operator=hop,
                                   measurement=(y_o.cuda(), yf_o.cuda()), guidance="I", device="cuda")
    hat = m(x.cuda(), sig.cuda()).cpu()
    assert float((hat - ref).abs().max()) < 1e-3


def test_error_behaviour(gold, tiny):
    import kdip_amd.condition as kc
    import kdip_amd.measurements as km
    models, D, sd, cfg = tiny
    hop, oop, (y, yf), x0 = make_ops("inpainting", gold)
    meas = (y.cuda(), yf.cuda())
    x = x0.cuda(); s = torch.tensor([1.0], device="cuda")
    mk = lambda **kw: kc.ConditionOpenAIDenoiser(inner_model=models["f32"], diffusion=D, recon_mse=None, operator=hop,
                                                 measurement=meas, device="cuda", **kw)
    with pytest.raises(ValueError):
        mk(guidance="bogus", x0_cov_type="convert")(x, s)
    with pytest.raises(AssertionError):
        mk(guidance="dps", x0_cov_type="dps")(x, s)
    with pytest.raises(ValueError):
        mk(guidance="I", x0_cov_type="nope")(x, s)
    with pytest.raises(NameError):
        km.get_operator("does_not_exist", device="cuda")
    with pytest.raises(NameError):
        km.register_operator("inpainting")(type("X", (), {}))

    class Dummy:
        name = "colorization"
    with pytest.raises(KeyError):
        kc.ConditionOpenAIDenoiser(inner_model=models["f32"], diffusion=D, x0_cov_type="convert", recon_mse=None,
                                   operator=Dummy(), measurement=meas, guidance="I")


# --------------------------------------------------------------------------- sampler ----
# Sampler-run bounds per arithmetic mode: (max-abs of the final image, |PSNR_hip - PSNR_ref| in dB against the ground truth).
#   f32     exact-f32 MFMA: the north_star tolerance (1e-3 dB) and a max-abs bound.
#   bf16x3  split-precision convs (per-conv error 5e-6 vs 1e-6, per guided call 3e-5 vs 3e-6 max-abs): held to the same 1e-3 dB.  No
#           max-abs bound: the tiny random-weight model saturates its output at +-1 and the 4-step trajectory is chaotic, so the
#           8 x larger per-call round-off can move single pixels by O(1) (one pixel of 64 x 64 x 3 flipping -1 -> +1 = 1.2e-3 dB);
#           the number of such pixels is printed and bounded at 1 % (it varies run to run on the churned Heun case: 1 ... 18 of 12 288).
#   bf16    production throughput mode: 3 x the measured deviation (0.004 / 0.014 dB).
SAMPLER_BOUNDS = {"f32": (5e-3, 1e-3), "bf16x3": (None, 1e-3), "f16x3": (None, 1e-3), "bf16": (None, 0.05)}      # (f16x3: the opt-in fp16-headed split, held to bf16x3's bounds)


def _sampler_case(models, D, hop, meas, x0, dt, fn, xT, sig, ref, **kw):
    import kdip_amd.condition as kc
    from kdip_amd.evaluation import psnr
    m = kc.ConditionOpenAIDenoiser(inner_model=models[dt], diffusion=D, x0_cov_type="convert", recon_mse=None,
                                   operator=hop, measurement=meas, guidance="I", device="cuda").eval()
    seen = []
    x = fn(m, xT.cuda(), sig, disable=True, callback=lambda d: seen.append(d["i"]), **kw).cpu()
    assert seen == [0, 1, 2, 3]
    err = float((x - ref).abs().max())
    dp = abs(float(psnr(x, x0)) - float(psnr(ref, x0)))
    moved = int(((x - ref).abs() > 1e-2).sum())
    return err, dp, moved


def test_sampler_golden(gold, tiny):
    import kdip_amd.sampling as ks
    models, D, sd, cfg = tiny
    g = gold("sampler")
    hop, oop, (y, yf), x0 = make_ops("gaussian_blur", gold)
    meas = (y.cuda(), yf.cuda())
    sig = ks.get_sigmas_karras(4, 0.01, 80, rho=7.0, device="cuda")
    assert torch.equal(sig.cpu(), T(g["sigmas"]))
    assert torch.equal(ks.get_sigmas_karras(100, 0.01, 80).cpu(), T(gold("tables")["sigmas100"]))
    bad = []
    for sampler, fn in (("heun", ks.sample_heun), ("euler", ks.sample_euler)):
        for dt, (emax, dpmax) in SAMPLER_BOUNDS.items():
            err, dp, moved = _sampler_case(models, D, hop, meas, x0, dt, fn, T(g["xT"]), sig, T(g[f"{sampler}.x0"]))
            print(f"\n{sampler} 4-step ode, tiny model, {dt}: max-abs {err:.2e}, dPSNR {dp:.1e} dB, {moved} of 12288 values moved > 1e-2 vs the reference capture")
            if (emax is not None and err >= emax) or dp >= dpmax or (dt in ("bf16x3", "f16x3") and moved > 122):
                bad.append((sampler, dt, err, dp, moved))
    assert not bad, bad


def test_sampler_churn_golden(gold, tiny):
    """Stochastic samplers (s_churn > 0, k_diffusion/sampling.py:123-127,164-169) against the reference's churn
    trajectories: the per-step noise is drawn on the CPU generator exactly as the reference capture drew it
    (torch.manual_seed(7), one randn_like per step) and injected through `noise_fn`, so kdip_sampler_add_noise is
    compared value-for-value."""
    import kdip_amd.sampling as ks
    models, D, sd, cfg = tiny
    g = gold("sampler")
    hop, oop, (y, yf), x0 = make_ops("gaussian_blur", gold)
    meas = (y.cuda(), yf.cuda())
    sig = ks.get_sigmas_karras(4, 0.01, 80, rho=7.0, device="cuda")
    bad = []
    for sampler, fn in (("heun", ks.sample_heun), ("euler", ks.sample_euler)):
        ref = T(g[f"{sampler}.x0_churn"])
        # the churned trajectory must differ from the ode one (the noise really went in)
        assert float((ref - T(g[f"{sampler}.x0"])).abs().max()) > 1e-3
        for dt in ("f32", "bf16x3", "f16x3"):
            emax, dpmax = SAMPLER_BOUNDS[dt]
            torch.manual_seed(7)
            cpu_noise = lambda x: torch.randn(x.shape)           # the reference's global CPU stream
            err, dp, moved = _sampler_case(models, D, hop, meas, x0, dt, fn, T(g["xT"]), sig, ref, s_churn=80, s_tmin=0.05, s_tmax=50,
                                           s_noise=1.003, noise_fn=cpu_noise)
            print(f"\nchurn {sampler} {dt}: max-abs {err:.2e}, dPSNR {dp:.2e} dB, {moved} of 12288 values moved > 1e-2")
            if (emax is not None and err >= emax) or dp >= dpmax or (dt in ("bf16x3", "f16x3") and moved > 122):
                bad.append((sampler, dt, err, dp, moved))
    assert not bad, bad


def test_sampler_dpmpp2m_golden(gold, tiny):
    """sample_dpmpp_2m on the HIP path (f32 mode) against the reference capture."""
    import kdip_amd.condition as kc
    import kdip_amd.sampling as ks
    models, D, sd, cfg = tiny
    g = gold("sampler_dpmpp2m")
    hop, oop, (y, yf), x0 = make_ops("gaussian_blur", gold)
    m = kc.ConditionOpenAIDenoiser(inner_model=models["f32"], diffusion=D, x0_cov_type="convert", recon_mse=None, operator=hop,
                                   measurement=(y.cuda(), yf.cuda()), guidance="I", device="cuda")
    first = []
    x = ks.sample_dpmpp_2m(m, T(g["xT"]).cuda(), T(g["sigmas"]), disable=True, callback=lambda d: first.append(d["denoised"].cpu()))
    assert float((first[0] - T(g["denoised_first"])).abs().max()) < 2e-3
    assert float((x.cpu() - T(g["x0"])).abs().max()) < 5e-3


def test_tmpd_stsl_golden(gold, tiny):
    """tmpd covariance (extra all-ones VJP) and STSL (one forward + VJP per Hutchinson probe) against the
    reference captures; probes are the same CPU-seeded eps the reference drew."""
    import kdip_amd.condition as kc
    models, D, sd, cfg = tiny
    g = gold("guided_calls_extra")
    modes = [("I", "tmpd", {}), ("II", "tmpd", {}), ("stsl", "dps", dict(zeta=1.0, eta=0.5, num_hutchinson_samples=2)),
             ("stsl+mle", "convert", dict(zeta=1.0, eta=0.5, num_hutchinson_samples=2))]
    for name in ("gaussian_blur", "inpainting"):
        hop, oop, (y, yf), x0 = make_ops(name, gold)
        meas = (y.cuda(), yf.cuda())
        for guidance, cov, extra in modes:
            for sigma_v in (1.5, 0.12):
                key = f"{name}|{guidance}|{cov}|{sigma_v}"
                if key not in g.files:
                    continue
                x = x0 + sigma_v * torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(11))
                m = kc.ConditionOpenAIDenoiser(inner_model=models["f32"], diffusion=D, x0_cov_type=cov, recon_mse=None,
                                               operator=hop, measurement=meas, guidance=guidance, zeta=extra.get("zeta"),
                                               eta=extra.get("eta"), num_hutchinson_samples=extra.get("num_hutchinson_samples"),
                                               mle_sigma_thres=0.2, device="cuda")
                torch.manual_seed(5)
                m.stsl_eps = [torch.randn_like(x) for _ in range(extra.get("num_hutchinson_samples") or 0)]
                hat = m(x.cuda(), torch.tensor([sigma_v], device="cuda")).cpu()
                err = float((hat - T(g[key])).abs().max())
                assert err < 3e-3, (key, err)


def test_analytic_variance_estimator(gold, tiny):
    import kdip_amd.analytic_variance as kav
    from helpers import smooth_image
    models, D, sd, cfg = tiny
    ga = gold("analytic_variance")
    sig = T(ga["sigmas"])
    batches = [smooth_image(2, 64, 21), smooth_image(2, 64, 22)]
    # same noise as the reference capture: drawn on the CPU stream in (sigma, batch) order
    torch.manual_seed(9)
    noises = [[torch.randn_like(b) for b in batches] for _ in sig]
    out = []
    import kdip_amd.external as ke
    from kdip_amd.sampling import _sigma_vec
    den = ke.OpenAIDenoiser(models["f32"], D)
    for i, s in enumerate(sig):
        acc = 0.0
        for j, b in enumerate(batches):
            xt = (b + noises[i][j] * s).cuda()
            hat = den(xt, _sigma_vec(xt, s)) if float(s) > 0 else xt
            acc += float((b.cuda() - hat).pow(2).mean())
        out.append(acc / len(batches))
    assert float((torch.tensor(out) - T(ga["mse_list"])).abs().max()) < 1e-4
    res = kav.estimate_recon_mse(models["f32"], D, [b.cuda() for b in batches], sigmas=sig)
    assert res["mse_list"].shape == (6,) and torch.isfinite(res["mse_list"]).all() and res["errors"].shape == (6, 2)


@pytest.mark.parametrize("ortho", ["dwt", "dct"])
def test_dwtvar_loss_value(tiny, ortho):
    """OpenAIDenoiserV2.loss (k_diffusion/external.py:145-159), forward value: HIP path (f32 mode) against the formula restated on the
    oracle UNet + oracle OrthoTransform, batch of 3 with two distinct sigmas."""
    import torch.nn.functional as F
    import kdip_amd.external as ke
    from oracle import unet as ounet
    from oracle.tables import DiffusionTables
    from oracle.transforms import OrthoTransform as OOT
    models, D, sd, cfg = tiny
    g = torch.Generator().manual_seed(4)
    x = (torch.rand(3, 3, 64, 64, generator=g) * 2 - 1) * 0.7
    noise = torch.randn(3, 3, 64, 64, generator=g)
    sigma = torch.tensor([0.5, 2.0, 0.5])
    den = ke.OpenAIDenoiserV2(models["f32"], D, ortho_tf_type=ortho)
    got = den.loss(x.cuda(), noise.cuda(), sigma.cuda()).cpu()
    T_, ot = DiffusionTables(), OOT(ortho)
    ref = []
    for i in range(3):
        s = sigma[i:i + 1]
        c_in, c_out = 1 / (s ** 2 + 1) ** 0.5, -s
        xn = x[i:i + 1] + noise[i:i + 1] * s
        out, feat = ounet.unet_forward(sd, cfg, xn * c_in, T_.sigma_to_t(s), return_feature=True)
        mo = out.chunk(2, dim=1)[0]
        logvar, logvar_ot = F.conv2d(feat, sd["out_cov.weight"], sd["out_cov.bias"]).chunk(2, dim=1)
        target = (x[i:i + 1] - xn) / c_out
        loss = (mo - target).pow(2) / logvar.exp() + logvar + (ot(mo) - ot(target)).pow(2) / logvar_ot.exp() + logvar_ot
        ref.append(float(loss.flatten(1).mean(1)))
    ref = torch.tensor(ref)
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-4, (got, ref)


@pytest.mark.gpu
def test_eps_mse_loss_value(tiny):
    """OpenAIDenoiser.loss = DiscreteEpsDDPMDenoiser.loss (k_diffusion/external.py:105-109): per-sample mean (eps_hat - noise)^2,
    HIP path (f32 mode) against the formula on the oracle UNet, batch of 3 with two distinct sigmas."""
    import kdip_amd.external as ke
    from oracle import unet as ounet
    from oracle.tables import DiffusionTables
    models, D, sd, cfg = tiny
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(3, 3, 64, 64, generator=g) * 2 - 1) * 0.7
    noise = torch.randn(3, 3, 64, 64, generator=g)
    sigma = torch.tensor([0.5, 2.0, 0.5])
    den = ke.OpenAIDenoiser(models["f32"], D)
    got = den.loss(x.cuda(), noise.cuda(), sigma.cuda()).cpu()
    T_ = DiffusionTables()
    ref = []
    for i in range(3):
        s = sigma[i:i + 1]
        c_in = 1 / (s ** 2 + 1) ** 0.5
        xn = x[i:i + 1] + noise[i:i + 1] * s
        eps = ounet.unet_forward(sd, cfg, xn * c_in, T_.sigma_to_t(s)).chunk(2, dim=1)[0]
        ref.append(float((eps - noise[i:i + 1]).pow(2).flatten(1).mean(1)))
    ref = torch.tensor(ref)
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-4, (got, ref)


@pytest.mark.parametrize("dt", ["f32", "bf16x3", "f16x3"])      # (f16x3: the replay's fp16-window flag is polled outside the capture; bf16x3: the cotangent-amax reduction of the VJP -- memset + kernel -- is captured too)
def test_hipgraph_capture_replay(gold, tiny, dt):
    """Capture / replay of guided calls (kdip_amd.graphs.GraphedDenoiser): the closed-form branch is captured once per sigma into a
    hipGraph and replayed for new inputs with results equal to the eager call (fp64-atomic order noise only); with capture_cg=False
    the CG branch stays eager (it reads convergence flags back to the host; test_hipgraph_cg_branch_fixed_trips covers its capture)."""
    import kdip_amd.condition as kc
    from kdip_amd.graphs import GraphedDenoiser
    models, D, sd, cfg = tiny
    hop, oop, (y, yf), x0 = make_ops("gaussian_blur", gold)
    den = kc.ConditionOpenAIDenoiser(inner_model=models[dt], diffusion=D, x0_cov_type="convert", recon_mse=None, operator=hop,
                                     measurement=(y.cuda(), yf.cuda()), guidance="I", device="cuda")     # (f32 mode: run-to-run noise is ~1e-6;
    gd = GraphedDenoiser(den, capture_cg=False)                                                          #  bf16 re-rounds the atomics-order noise to ~1e-3)
    g = torch.Generator().manual_seed(21)
    xs = [(x0 + 1.5 * torch.randn(1, 3, 64, 64, generator=g)).cuda() for _ in range(3)]
    sig = torch.tensor([1.5], device="cuda")
    outs = [gd(x, sig) for x in xs]                      # call 0 captures (+ replays), calls 1, 2 replay
    assert gd.replays == 3 and gd.eager_calls == 0 and len(gd._graphs) == 1
    for x, o in zip(xs, outs):
        ref = den(x, sig)
        assert float((o - ref).abs().max()) < 1e-4
    assert float((outs[0] - outs[1]).abs().max()) > 1e-3          # different inputs really went through
    lo = torch.tensor([0.12], device="cuda")                      # CG branch: eager
    o = gd(xs[0], lo)
    assert gd.eager_calls == 1 and float((o - den(xs[0], lo)).abs().max()) < 1e-4


def test_hipgraph_invalidated_by_workspace_regrowth(gold):
    """A call at a larger batch shape makes the UNet handle free + re-allocate its workspace arenas (unet.hip ensure_workspace): a
    graph captured before that holds dangling pointers.  GraphedDenoiser records the handle's workspace generation and the
    measurement tensors it captured against, drops its graphs when either changes, and re-captures on the next use."""
    import kdip_amd.condition as kc
    import kdip_amd.unet as ku
    from kdip_amd.graphs import GraphedDenoiser
    from oracle import unet as ounet
    cfg = ounet.UNetConfig(**ounet.TINY)
    sd = ounet.init_state_dict(cfg, seed=0)
    m = ku.UNetModel(image_size=64, model_channels=32, num_res_blocks=1, attention_resolutions="32", channel_mult=(1, 2), dtype="f32", device="cuda")
    m.load_state_dict(sd)                                 # a fresh handle: its arenas start empty and grow with the batch
    D = ku.GaussianDiffusionTables()
    hop, oop, (y, yf), x0 = make_ops("gaussian_blur", gold)
    den = kc.ConditionOpenAIDenoiser(inner_model=m, diffusion=D, x0_cov_type="convert", recon_mse=None, operator=hop,
                                     measurement=(y.cuda(), yf.cuda()), guidance="I", device="cuda")
    gd = GraphedDenoiser(den)
    g = torch.Generator().manual_seed(22)
    x1 = (x0 + 1.5 * torch.randn(1, 3, 64, 64, generator=g)).cuda()
    x4 = (x0 + 1.5 * torch.randn(4, 3, 64, 64, generator=g)).cuda()
    s1, s4 = torch.tensor([1.5], device="cuda"), torch.full((4,), 1.5, device="cuda")
    o1 = gd(x1, s1).clone()
    gen0 = m.workspace_generation()
    o4 = gd(x4, s4)                                       # larger batch: arenas regrown, the batch-1 graph is stale
    assert m.workspace_generation() > gen0 and gd.invalidations >= 1
    assert all(k[1][0] == 4 for k in gd._graphs)          # only graphs captured against the new arenas survive
    o1b = gd(x1, s1)                                      # re-captured, not replayed from freed memory
    assert float((o1b - o1).abs().max()) < 1e-4
    assert float((o4 - den(x4, s4)).abs().max()) < 1e-4
    assert float((gd(x1, s1) - den(x1, s1)).abs().max()) < 1e-4


@pytest.mark.parametrize("sigma_v", [1.5, 0.12])
def test_fused_guided_call_equals_stepwise(gold, tiny, sigma_v):
    """kdip_guided_call_v1 (one C entry point per Type-I guided call) = the stepwise path (uncond_pred -> solver -> cotangent -> VJP ->
    combine through ~10 C calls): same kernels, same order; closed-form branch and CG branch (adaptive and fixed-trip)."""
    import kdip_amd.condition as kc
    models, D, sd, cfg = tiny
    hop, oop, (y, yf), x0 = make_ops("gaussian_blur", gold)
    g = torch.Generator().manual_seed(31)
    x = (x0 + sigma_v * torch.randn(2, 3, 64, 64, generator=g)).cuda()
    sig = torch.full((2,), sigma_v, device="cuda")
    outs = {}
    for fused in (False, True):
        den = kc.ConditionOpenAIDenoiser(inner_model=models["f32"], diffusion=D, x0_cov_type="convert", recon_mse=None, operator=hop,
                                         measurement=(y.cuda(), yf.cuda()), guidance="I", device="cuda")
        den.fused_call = fused
        outs[fused] = den(x, sig)
        iters = list(hop.cg_iters)
    assert float((outs[True] - outs[False]).abs().max()) < 2e-5          # fp64-atomic order noise of two runs of the same kernels
    if sigma_v < 0.2:
        assert max(iters) > 0
        hop.set_cg_fixed_trips(int(1.5 * max(iters)) + 4)                 # capture-safe mode: no host read, surplus iterations are no-ops
        try:
            fixed = den(x, sig)
            assert hop.cg_iters == [-1, -1] and hop.cg_unconverged() == 0
            assert float((fixed - outs[True]).abs().max()) < 2e-5
            hop.set_cg_fixed_trips(1)                                     # too few trips: flagged by the sticky counter
            den(x, sig)
            assert hop.cg_unconverged() == 1 and hop.cg_unconverged() == 0
        finally:
            hop.set_cg_fixed_trips(0)


def test_hipgraph_cg_branch_fixed_trips(gold, tiny):
    """The CG branch captured as a fixed-trip solve: replay = eager adaptive call, no unconverged replay."""
    import kdip_amd.condition as kc
    from kdip_amd.graphs import GraphedDenoiser
    models, D, sd, cfg = tiny
    hop, oop, (y, yf), x0 = make_ops("gaussian_blur", gold)
    den = kc.ConditionOpenAIDenoiser(inner_model=models["f32"], diffusion=D, x0_cov_type="convert", recon_mse=None, operator=hop,
                                     measurement=(y.cuda(), yf.cuda()), guidance="I", device="cuda")
    gd = GraphedDenoiser(den)
    g = torch.Generator().manual_seed(23)
    xs = [(x0 + 0.12 * torch.randn(1, 3, 64, 64, generator=g)).cuda() for _ in range(3)]
    lo = torch.tensor([0.12], device="cuda")
    outs = [gd(x, lo) for x in xs]
    assert gd.replays == 3 and gd.eager_calls == 0 and len(gd.cg_trips) == 1
    assert gd.cg_unconverged() == 0
    for x, o in zip(xs, outs):
        assert float((o - den(x, lo)).abs().max()) < 1e-4


def test_graph_redoes_unconverged_fixed_trip_replay(gold, tiny):
    """a replay whose captured trip count is too small for its input must not be returned silently.  The graph is
    captured with a deliberately tiny margin (cg_margin 0 -> 4 trips); the wrapper polls the sticky counter after the replay,
    warns, redoes the call with the adaptive solver (result = eager) and re-captures the key with twice the trips."""
    import warnings
    import kdip_amd.condition as kc
    from kdip_amd.graphs import GraphedDenoiser
    models, D, sd, cfg = tiny
    hop, oop, (y, yf), x0 = make_ops("gaussian_blur", gold)
    den = kc.ConditionOpenAIDenoiser(inner_model=models["f32"], diffusion=D, x0_cov_type="convert", recon_mse=None, operator=hop,
                                     measurement=(y.cuda(), yf.cuda()), guidance="I", device="cuda")
    gd = GraphedDenoiser(den, cg_margin=0.0)
    x = (x0 + 0.12 * torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(29))).cuda()
    lo = torch.tensor([0.12], device="cuda")
    ref = den(x, lo)
    need = max(hop.cg_iters)
    assert need > 4, need                      # (otherwise 4 trips would be enough and the test would be vacuous)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = gd(x, lo)
    assert any("unconverged" in str(m.message) for m in w)
    assert gd.cg_redone == 1 and list(gd.cg_trips.values()) == [8]
    assert float((out - ref).abs().max()) < 1e-4
    assert gd.cg_unconverged() == 0            # nothing pending after the redo


def test_second_vjp_after_fused_call_uses_exported_layout(gold, tiny):
    """x0_raw of a fused Type-I call is found through kdip_guided_ws_layout (no hard-coded slice), so a second
    VJP right after it on the same denoiser -- what tmpd / STSL do -- equals the stepwise path's."""
    import ctypes as C
    import kdip_amd._lib as L
    import kdip_amd.condition as kc
    models, D, sd, cfg = tiny
    hop, oop, (y, yf), x0 = make_ops("gaussian_blur", gold)
    mk = lambda: kc.ConditionOpenAIDenoiser(inner_model=models["f32"], diffusion=D, x0_cov_type="convert", recon_mse=None, operator=hop,
                                            measurement=(y.cuda(), yf.cuda()), guidance="I", device="cuda")
    off = (C.c_long * L.GWS_COUNT)()
    L.check(L.load().kdip_guided_ws_layout(2, 64, off, L.GWS_COUNT))
    n3 = 3 * 2 * 64 * 64
    assert list(off) == sorted(off) and off[L.GWS_X0_RAW] - off[L.GWS_X0_MEAN] == n3 and off[L.GWS_SCORE] + n3 <= L.load().kdip_guided_ws_floats(2, 64)
    with pytest.raises(L.KdipError):
        L.check(L.load().kdip_guided_ws_layout(2, 64, off, L.GWS_COUNT - 1))
    x = (x0 + 1.5 * torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(31))).cuda()
    s = torch.full((2,), 1.5, device="cuda")
    gh = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(32)).cuda()
    fused, step = mk(), mk()
    step.fused_call = False
    hf, hs = fused(x, s), step(x, s)
    assert float((hf - hs).abs().max()) < 2e-5
    v1, v2 = fused._vjp_x0(gh), step._vjp_x0(gh)                     # the second VJP (all-ones for tmpd, Hutchinson probes for STSL)
    assert float((v1 - v2).abs().max()) < 2e-5 * max(1.0, float(v2.abs().max()))
    # tmpd / stsl denoisers on the same UNet handle right after a fused call: finite, and equal to a fresh denoiser's answer
    for guidance, cov, kw in (("I", "tmpd", {}), ("stsl", "dps", dict(zeta=1.0, eta=0.5, num_hutchinson_samples=1))):
        a = kc.ConditionOpenAIDenoiser(inner_model=models["f32"], diffusion=D, x0_cov_type=cov, recon_mse=None, operator=hop,
                                       measurement=(y.cuda(), yf.cuda()), guidance=guidance, device="cuda", **kw)
        fused(x, s)
        torch.manual_seed(5); o1 = a(x, s)
        torch.manual_seed(5); o2 = a(x, s)
        assert torch.isfinite(o1).all() and float((o1 - o2).abs().max()) < 1e-4
    # a wrong spatial size is refused before any workspace is sized from it
    with pytest.raises(ValueError):
        fused(torch.zeros(1, 3, 32, 32, device="cuda"), torch.tensor([1.5], device="cuda"))


def test_custom_mat_solver_is_honoured_by_type_I(gold, tiny):
    """a solver registered through the public register_mat_solver surface (condition/condition.py:307-314) must be the
    one Type-I uses -- the fused entry point is only taken for the library's own solvers."""
    import kdip_amd.condition as kc
    models, D, sd, cfg = tiny
    hop, oop, (y, yf), x0 = make_ops("gaussian_blur", gold)
    calls = []
    builtin = kc.__MAT_SOLVER__["gaussian_blur"]

    @kc.register_mat_solver("gaussian_blur")
    def half_mat(operator, y_, x0_mean, theta0_var, ortho_tf=None):
        calls.append(1)
        return 0.5 * builtin(operator, y_, x0_mean, theta0_var, ortho_tf)
    try:
        den = kc.ConditionOpenAIDenoiser(inner_model=models["f32"], diffusion=D, x0_cov_type="analytic", recon_mse=synthetic_recon_mse(),
                                         operator=hop, measurement=(y.cuda(), yf.cuda()), guidance="I", device="cuda")
        x = (x0 + 1.5 * torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(33))).cuda()
        s = torch.tensor([1.5], device="cuda")
        out = den(x, s)
        assert calls, "the registered solver was bypassed"
    finally:
        kc.__MAT_SOLVER__["gaussian_blur"] = builtin
    ref = kc.ConditionOpenAIDenoiser(inner_model=models["f32"], diffusion=D, x0_cov_type="analytic", recon_mse=synthetic_recon_mse(),
                                     operator=hop, measurement=(y.cuda(), yf.cuda()), guidance="I", device="cuda")(x, s)
    assert float((out - ref).abs().max()) > 1e-4          # half the likelihood score: a different estimate
