"""GPU: full-size (256x256) parity of one guided call per BASELINE model against the CPU oracle,
and finite/PSNR sanity of a short sampler run for every BASELINE config shape.  Sizes are chosen so
the oracle side finishes in seconds (batch 1, one or two calls)."""
import numpy as np
import pytest
import torch

from helpers import smooth_image, psnr_db

pytestmark = pytest.mark.gpu


def _setup(cfg_name, op_name, dtype, B=1, out_cov=False):
    import kdip_amd.unet as ku
    import kdip_amd.measurements as km
    from oracle import unet as ounet, operators as oops
    ocfg = ounet.UNetConfig(**getattr(ounet, cfg_name))
    sd = ounet.init_state_dict(ocfg, seed=0, out_cov=out_cov)
    kcfg = ku.FFHQ_CONFIG if cfg_name == "FFHQ" else ku.IMAGENET_CONFIG
    m = ku.UNetModel(dtype=dtype, **kcfg)
    m.load_state_dict(sd)
    opkw = {"gaussian_blur": dict(in_shape=(1, 3, 256, 256), kernel_size=61, intensity=3.0, sigma_s=0.05),
            "motion_blur": dict(in_shape=(1, 3, 256, 256), kernel_size=61, intensity=0.5, sigma_s=0.05),
            "super_resolution": dict(in_shape=(1, 3, 256, 256), scale_factor=4, sigma_s=0.05),
            "inpainting": dict(sigma_s=0.05, mask_opt=dict(mask_type="random", mask_prob_range=(0.5, 0.5), image_size=256))}[op_name]
    np.random.seed(0)
    hop = km.get_operator(op_name, device="cuda", **opkw)
    np.random.seed(0)
    oop = oops.get_operator(op_name, **opkw)
    x0 = smooth_image(B, 256, 1)
    torch.manual_seed(2)
    meas = oop.forward(x0.clone(), flatten=True)
    return m, sd, ocfg, hop, oop, meas, x0


def _oracle_admissible(oden, hip_raw, hat, x, sigma):
    """The oracle's answer(s) for one guided call, and the distance of the HIP output `hat` to the nearest of them.

    The guided output is DISCONTINUOUS where |x0_raw| crosses 1 (VJP through `pred_xstart.clamp(-1, 1)`, condition.py:231,
    gaussian_diffusion.py:293-311): a pixel with |x0_raw| = 1 to within rounding has two correct answers.  The oracle first answers
    with ITS OWN clamp mask -- nothing of the HIP run reaches it.  Only if the HIP run's mask differs is a second oracle answer
    computed, with the oracle's mask flipped at exactly the differing pixels; the oracle itself refuses any pixel that is not within
    1e-4 of the boundary in its own x0_raw (oracle/condition.py: clamp_flip), and at most 4 pixels may differ.  The HIP output must
    equal one of the two answers.  Returns (reference nearest to hat, max-abs error against it, [(pixel, oracle x0_raw)] flipped)."""
    oden.clamp_flip = None
    memo = getattr(oden, "_own_memo", None)      # a test that checks several arithmetic modes against the same oracle call pays for it once
    if memo is not None and torch.equal(memo[0], x) and torch.equal(memo[1], sigma):
        ref_own, own_raw, oden.last_borderline = memo[2], memo[3], memo[4]
    else:
        ref_own = oden(x, sigma)
        own_raw = oden.last_x0_raw.clone()
        oden._own_memo = (x.clone(), sigma.clone(), ref_own, own_raw, list(oden.last_borderline))
    err_own = float((hat - ref_own).abs().max())
    differ = (hip_raw.abs() <= 1) != (own_raw.abs() <= 1)
    idx = [tuple(i) for i in differ.nonzero().tolist()]
    if not idx:
        return ref_own, err_own, []
    assert len(idx) <= 4, idx
    assert set(idx) <= set(oden.last_borderline), (idx, oden.last_borderline)
    oden.clamp_flip = idx
    try:
        ref_flip = oden(x, sigma)
    finally:
        oden.clamp_flip = None
    err_flip = float((hat - ref_flip).abs().max())
    flips = [(i, float(own_raw[i])) for i in idx]
    return (ref_flip, err_flip, flips) if err_flip <= err_own else (ref_own, err_own, flips)


@pytest.mark.parametrize("sigma_v", [1.5, 0.12])
def test_ffhq_type1_convert_fullsize(sigma_v):
    """BASELINE configs[1] -- the benchmarked configuration -- at 256 x 256, batch 1, one closed-form call (sigma 1.5) and one CG-branch
    call (sigma 0.12) against the CPU oracle: f32 AND bf16x3 (the headline arithmetic of bench.py) within 2e-4 max-abs; bf16 reported as
    PSNR with a measured floor.  (condition/condition.py:167-174,231-248,351-386)"""
    import kdip_amd.unet as ku
    import kdip_amd.condition as kc
    from oracle import condition as ocond
    m, sd, ocfg, hop, oop, meas, x0 = _setup("FFHQ", "gaussian_blur", "f32")
    x = x0 + sigma_v * torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(11))
    D = ku.GaussianDiffusionTables()
    measd = (meas[0].cuda(), meas[1].cuda())
    oden = ocond.GuidedDenoiser(sd, ocfg, oop, meas, "I", x0_cov_type="convert")
    errs, ref32 = {}, None
    for dtype in ("f32", "bf16x3", "f16x3"):
        if dtype != "f32":
            del m, hm
            torch.cuda.empty_cache()
            m = ku.UNetModel(dtype=dtype, **ku.FFHQ_CONFIG); m.load_state_dict(sd)
        hm = kc.ConditionOpenAIDenoiser(inner_model=m, diffusion=D, x0_cov_type="convert", recon_mse=None, operator=hop,
                                        measurement=measd, guidance="I", device="cuda")
        hat = hm(x.cuda(), torch.tensor([sigma_v], device="cuda")).cpu()
        ref, errs[dtype], flips = _oracle_admissible(oden, hm._stash[0].cpu(), hat, x, torch.tensor([sigma_v]))
        print(f"\nFFHQ configs[1] full-size sigma={sigma_v} {dtype}: max-abs {errs[dtype]:.2e}; borderline clamp pixels {flips}")
        if dtype == "f32":
            ref32 = ref
    assert errs["f32"] < 2e-4 and errs["bf16x3"] < 2e-4 and errs["f16x3"] < 2e-4, errs          # measured <= 7.6e-5
    assert m.x3_fallbacks == 0                       # (the last handle is the f16x3 one: its own arithmetic produced the result)
    del m, hm
    torch.cuda.empty_cache()
    m2 = ku.UNetModel(dtype="bf16", **ku.FFHQ_CONFIG); m2.load_state_dict(sd)
    hm2 = kc.ConditionOpenAIDenoiser(inner_model=m2, diffusion=D, x0_cov_type="convert", recon_mse=None, operator=hop,
                                     measurement=measd, guidance="I", device="cuda")
    hat2 = hm2(x.cuda(), torch.tensor([sigma_v], device="cuda")).cpu()
    p = psnr_db(hat2, ref32)
    print(f"FFHQ full-size sigma={sigma_v}: bf16 PSNR(hip, oracle) {p:.1f} dB")
    assert p > (30.0 if sigma_v > 1 else 64.0)      # measured 35.2 / 69.5 dB: floor = measured - 5 dB


def test_imagenet_unet_fullsize():
    """ImageNet-256 architecture (552.8 M params, attention at 32/16/8, T up to 1024): UNet forward and
    input-VJP in f32 mode against oracle autograd at batch 1."""
    from oracle import unet as ounet
    m, sd, ocfg, hop, oop, meas, x0 = _setup("IMAGENET", "motion_blur", "f32")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 3, 256, 256, generator=g)
    t = torch.tensor([321.0])
    cot = torch.randn(1, 6, 256, 256, generator=g)
    xo = x.clone().requires_grad_()
    out_ref = ounet.unet_forward(sd, ocfg, xo, t)
    vjp_ref = torch.autograd.grad((out_ref * cot).sum(), xo)[0]
    out, _, _ = m.forward_raw(x.cuda(), t.cuda())
    vjp = m.vjp(cot.cuda())
    e1 = float((out.cpu() - out_ref.detach()).abs().max() / out_ref.detach().abs().max())
    e2 = float((vjp.cpu() - vjp_ref).abs().max() / vjp_ref.abs().max())
    print(f"\nImageNet-256 UNet f32: fwd rel err {e1:.2e}, vjp rel err {e2:.2e}")
    assert e1 < 5e-4 and e2 < 5e-4
    # the split-precision mode on the same architecture (256 base channels, 4 - 16 heads, fp32 attention GEMMs): the f32 bound
    import kdip_amd.unet as ku
    del m
    torch.cuda.empty_cache()
    m3 = ku.UNetModel(dtype="bf16x3", **ku.IMAGENET_CONFIG); m3.load_state_dict(sd)
    out3, _, _ = m3.forward_raw(x.cuda(), t.cuda())
    vjp3 = m3.vjp(cot.cuda())
    e13 = float((out3.cpu() - out_ref.detach()).abs().max() / out_ref.detach().abs().max())
    e23 = float((vjp3.cpu() - vjp_ref).abs().max() / vjp_ref.abs().max())
    print(f"ImageNet-256 UNet bf16x3: fwd rel err {e13:.2e}, vjp rel err {e23:.2e}")
    assert e13 < 5e-4 and e23 < 5e-4
    # ... and the opt-in fp16-headed split (whatever the window watch decides for this input: the result holds the same bound)
    del m3
    torch.cuda.empty_cache()
    mh = ku.UNetModel(dtype="f16x3", **ku.IMAGENET_CONFIG); mh.load_state_dict(sd)
    outh, _, _ = mh.forward_raw(x.cuda(), t.cuda())
    vjph = mh.vjp(cot.cuda())
    e1h = float((outh.cpu() - out_ref.detach()).abs().max() / out_ref.detach().abs().max())
    e2h = float((vjph.cpu() - vjp_ref).abs().max() / vjp_ref.abs().max())
    print(f"ImageNet-256 UNet f16x3: fwd rel err {e1h:.2e}, vjp rel err {e2h:.2e} ({mh.x3_fallbacks} passes redone bf16-headed)")
    assert e1h < 5e-4 and e2h < 5e-4
    # ... with one fp16 window per dgrad launch (what bench.py --workload cfg3 runs in f16x3): the same bound, bitwise repeatable, and no pass flagged
    mh.set_x3_window("launch")
    nfb = mh.x3_fallbacks
    mh.forward_raw(x.cuda(), t.cuda())
    vl = mh.vjp(cot.cuda())
    mh.forward_raw(x.cuda(), t.cuda())
    vl2 = mh.vjp(cot.cuda())
    e2l = float((vl.cpu() - vjp_ref).abs().max() / vjp_ref.abs().max())
    print(f"ImageNet-256 UNet f16x3, per-launch windows: vjp rel err {e2l:.2e} ({mh.x3_fallbacks - nfb} passes redone)")
    assert e2l < 5e-4 and torch.equal(vl, vl2)


@pytest.mark.parametrize("sigma_v", [1.5, 0.12])
def test_imagenet_motion_typeI_analytic_fullsize(sigma_v):
    """BASELINE configs[3]: ImageNet-256 UNet + motion deblur + Type-I guidance + Analytic covariance
    (configs/test_imagenet.json:13-17, condition/condition.py:250-254, measurements.py:125-160) at batch 2,
    one high-sigma call (sigma^2/(1+sigma^2) branch) and one low-sigma call (table lookup branch), HIP f32 and bf16
    against the oracle at full size.  f32: <= 2e-3 max-abs; bf16: PSNR(hip, oracle) printed and bounded below."""
    import kdip_amd.unet as ku
    import kdip_amd.condition as kc
    from oracle import condition as ocond
    from helpers import synthetic_recon_mse
    m, sd, ocfg, hop, oop, meas, x0 = _setup("IMAGENET", "motion_blur", "f32", B=2)
    rm = synthetic_recon_mse()
    D = ku.GaussianDiffusionTables()
    measd = (meas[0].cuda(), meas[1].cuda())
    rmd = {k: v.cuda() for k, v in rm.items()}
    hm = kc.ConditionOpenAIDenoiser(inner_model=m, diffusion=D, x0_cov_type="analytic", recon_mse=rmd, operator=hop,
                                    measurement=measd, guidance="I", device="cuda")
    # (borderline |x0_raw| = 1 pixels have two correct answers: _oracle_admissible; with this input one pixel of image 0 has x0_raw = -1 +- 2e-6)
    x = x0 + sigma_v * torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(11))
    hat = hm(x.cuda(), torch.full((2,), sigma_v, device="cuda")).cpu()
    oden = ocond.GuidedDenoiser(sd, ocfg, oop, meas, "I", x0_cov_type="analytic", recon_mse=rm)
    ref, err, flips = _oracle_admissible(oden, hm._stash[0].cpu(), hat, x, torch.full((2,), sigma_v))
    del m, hm
    torch.cuda.empty_cache()
    m3 = ku.UNetModel(dtype="bf16x3", **ku.IMAGENET_CONFIG); m3.load_state_dict(sd)
    hm3 = kc.ConditionOpenAIDenoiser(inner_model=m3, diffusion=D, x0_cov_type="analytic", recon_mse=rmd, operator=hop,
                                     measurement=measd, guidance="I", device="cuda")
    hat3 = hm3(x.cuda(), torch.full((2,), sigma_v, device="cuda")).cpu()
    ref3, err3, flips3 = _oracle_admissible(oden, hm3._stash[0].cpu(), hat3, x, torch.full((2,), sigma_v))
    print(f"\nconfigs[3] sigma={sigma_v}: bf16x3 max-abs {err3:.2e} (borderline clamp pixels flipped: {flips3})")
    assert err3 < 2e-3, err3                 # the split-precision mode at the f32 bound
    del m3, hm3
    torch.cuda.empty_cache()
    m2 = ku.UNetModel(dtype="bf16", **ku.IMAGENET_CONFIG); m2.load_state_dict(sd)
    hm2 = kc.ConditionOpenAIDenoiser(inner_model=m2, diffusion=D, x0_cov_type="analytic", recon_mse=rmd, operator=hop,
                                     measurement=measd, guidance="I", device="cuda")
    hat2 = hm2(x.cuda(), torch.full((2,), sigma_v, device="cuda")).cpu()
    p = psnr_db(hat2, ref)
    print(f"\nconfigs[3] ImageNet motion Type-I analytic sigma={sigma_v} B=2 (borderline clamp pixels flipped: {flips}): f32 max-abs {err:.2e}; bf16 PSNR(hip, oracle) {p:.1f} dB, "
          f"bf16 max-abs {float((hat2 - ref).abs().max()):.2e}")
    assert err < 2e-3, err
    assert torch.isfinite(hat2).all() and p > (25.0 if sigma_v > 1 else 61.0)      # measured 30.4 dB (sigma 1.5: clamp flips on saturated random-weight outputs) / 66.9 dB (sigma 0.12); floor = measured - 5 dB


# bf16 PSNR(hip, oracle) floors of the three configs = measured (52.3 / 69.7, 53.2 / 70.2, 41.0 / 54.4 dB) minus 5 dB; key: (id, high sigma?)
BF16_FLOOR = {("cfg0_inpaint_dps", True): 47.0, ("cfg0_inpaint_dps", False): 64.5, ("cfg2_sr4_typeII_pgdm", True): 48.0, ("cfg2_sr4_typeII_pgdm", False): 65.0,
              ("cfg4_gauss_v2_dwt_autoI", True): 36.0, ("cfg4_gauss_v2_dwt_autoI", False): 49.0}
FULLSIZE = [
    # BASELINE configs[0], [2], [4] (configs[1] and [3] have their own tests above): (id, operator, guidance, cov, extra, v2 basis, low sigma)
    ("cfg0_inpaint_dps", "inpainting", "dps", "dps", dict(zeta=1.0), None, 0.12),                 # condition.py:140-148, measurements.py:202-244
    ("cfg2_sr4_typeII_pgdm", "super_resolution", "II", "pgdm", {}, None, 0.12),                   # condition.py:176-183, measurements.py:86-122
    ("cfg4_gauss_v2_dwt_autoI", "gaussian_blur", "autoI", None, {}, "dwt", 0.5),                  # condition.py:287-300,133-138; CG in the DWT basis below sigma 1
]


@pytest.mark.parametrize("cid,opn,guid,cov,extra,ortho,sig_lo", FULLSIZE)
def test_baseline_configs_fullsize_vs_oracle(cid, opn, guid, cov, extra, ortho, sig_lo):
    """One high-sigma and one low-sigma guided call of BASELINE configs[0] / [2] / [4] at 256 x 256 (FFHQ architecture, batch 2)
    against the CPU oracle on the same inputs: f32 and bf16x3 within 2e-3 max-abs (the f32-mode bound of the guided-call goldens),
    bf16 as PSNR(hip, oracle), printed and bounded.  configs[4] rides on the Haar-3 DWT layout of pywt
    (pinned against PyWavelets 1.1.1 since round 5: tests/test_thirdparty_pins.py)."""
    import kdip_amd.unet as ku
    import kdip_amd.condition as kc
    import kdip_amd.external as ke
    from oracle import condition as ocond
    B = 2
    m, sd, ocfg, hop, oop, meas, x0 = _setup("FFHQ", opn, "f32", B=B, out_cov=ortho is not None)
    D = ku.GaussianDiffusionTables()
    measd = (meas[0].cuda(), meas[1].cuda())

    def hip_den(model):
        if ortho is None:
            return kc.ConditionOpenAIDenoiser(inner_model=model, diffusion=D, x0_cov_type=cov, recon_mse=None, operator=hop, measurement=measd,
                                              guidance=guid, zeta=extra.get("zeta"), device="cuda").eval()
        return kc.ConditionOpenAIDenoiserV2(ke.OpenAIDenoiserV2(model, D, ortho_tf_type=ortho), operator=hop, measurement=measd, guidance=guid,
                                            mle_sigma_thres=1.0, device="cuda", ortho_tf_type=ortho).eval()
    outs = {}
    for dtype in ("f32", "bf16x3", "f16x3", "bf16"):
        if dtype != "f32":
            del m
            torch.cuda.empty_cache()
            m = ku.UNetModel(dtype=dtype, **ku.FFHQ_CONFIG); m.load_state_dict(sd)
        den = hip_den(m)
        for sigma_v in (1.5, sig_lo):
            x = x0 + sigma_v * torch.randn(B, 3, 256, 256, generator=torch.Generator().manual_seed(11))
            hat = den(x.cuda(), torch.full((B,), sigma_v, device="cuda")).cpu()
            raw = den._stash[0].cpu() if (ortho is None and guid != "II") else None      # V1 paths with a VJP: x0_raw for the clamp mask
            outs[(dtype, sigma_v)] = (x, hat, raw)
    for sigma_v in (1.5, sig_lo):
        x, hat, raw = outs[("f32", sigma_v)]
        if ortho is None:
            oden = ocond.GuidedDenoiser(sd, ocfg, oop, meas, guid, x0_cov_type=cov, zeta=extra.get("zeta"))
        else:
            oden = ocond.GuidedDenoiser(sd, ocfg, oop, meas, guid, mle_sigma_thres=1.0, v2=True, ortho_tf_type=ortho)
        flips = []
        if raw is not None:      # V1 paths with a VJP through the clamp: borderline |x0_raw| = 1 pixels have two correct answers (_oracle_admissible)
            ref, e32, flips = _oracle_admissible(oden, raw, hat, x, torch.full((B,), sigma_v))
            _, ex3, fl3 = _oracle_admissible(oden, outs[("bf16x3", sigma_v)][2], outs[("bf16x3", sigma_v)][1], x, torch.full((B,), sigma_v))
            _, eh3, flh = _oracle_admissible(oden, outs[("f16x3", sigma_v)][2], outs[("f16x3", sigma_v)][1], x, torch.full((B,), sigma_v))
            flips = flips + fl3 + flh
        else:
            ref = oden(x, torch.full((B,), sigma_v))
            e32 = float((hat - ref).abs().max())
            ex3 = float((outs[("bf16x3", sigma_v)][1] - ref).abs().max())
            eh3 = float((outs[("f16x3", sigma_v)][1] - ref).abs().max())
        p16 = psnr_db(outs[("bf16", sigma_v)][1], ref)
        iters = f", oracle CG iterations {oden.cg_stats.get('iters')}" if oden.cg_stats.get("iters") is not None else ""
        print(f"\n{cid} sigma={sigma_v} B={B} (borderline clamp pixels flipped: {flips}{iters}): f32 max-abs {e32:.2e}; bf16x3 max-abs {ex3:.2e}; f16x3 max-abs {eh3:.2e}; bf16 PSNR(hip, oracle) {p16:.1f} dB")
        assert e32 < 2e-4 and ex3 < 2e-4 and eh3 < 2e-4, (cid, sigma_v, e32, ex3, eh3)       # measured <= 6.7e-5 (f32) / 6.0e-5 (bf16x3)
        assert p16 > BF16_FLOOR[(cid, sigma_v > 1)], (cid, sigma_v, p16)


E2E = [
    # (operator, guidance, covariance, extra)  -- BASELINE configs[0..3] guidance per operator
    ("gaussian_blur", "I", "convert", {}),
    ("motion_blur", "I", "convert", {}),
    ("super_resolution", "II", "pgdm", {}),
    ("inpainting", "dps", "dps", dict(zeta=1.0)),
]


class _Teacher:
    """Records every model call of a sampler run: input x, sigma, the guided output and (V1 paths with a VJP) x0_raw."""

    def __init__(self, den):
        self.den, self.calls = den, []

    def __call__(self, x, sigma, **kw):
        out = self.den(x, sigma, **kw)
        raw = self.den._stash[0].clone() if self.den.guidance != "II" else None
        self.calls.append((x.clone(), float(getattr(sigma, "_kdip_host_value", sigma.reshape(-1)[0])), out.clone(), raw))
        return out


def _forced_mask_call(den, x, sigma, raw_teacher):
    """One guided call on the stepwise path with the TEACHER's clamp-gradient mask: wherever 1[|x0_raw| <= 1] of this run differs
    from the teacher's, the teacher's x0_raw is substituted in the stash the VJP cotangent reads (condition.py: _vjp_x0)."""
    orig, fused = den.uncond_pred, den.fused_call

    def patched(xx, ss):
        r = orig(xx, ss)
        st = list(den._stash)
        differ = (st[0].abs() <= 1) != (raw_teacher.abs() <= 1)
        st[0] = torch.where(differ, raw_teacher, st[0])
        den._stash = tuple(st)
        return r
    den.uncond_pred, den.fused_call = patched, False
    try:
        return den(x, sigma)
    finally:
        den.uncond_pred, den.fused_call = orig, fused


# Teacher-forced bounds.  A guided output is x0 + sigma^2 * (VJP of the UNet), clipped: an arithmetic error of the UNet is amplified by
# sigma^2 (6400 at the first call of the schedule) before the clip, so a per-call bound has to scale with it.
#   bf16x3: max-abs <= 1e-4 * max(2, sigma^2) on every call -- 2e-4 (the bound of the full-size oracle comparisons) up to sigma 1.4, and
#           3.5 x the measured 2.8e-5 * sigma^2 above (measured: 1.2e-7 at sigma 0.01 ... 1.7e-5 at 1.8 ... 7.8e-2 at 80; the parity modes are
#           deterministic, so these values reproduce bit for bit on one build) -- and per-call PSNR(bf16x3, f32) > 65 dB (measured >= 69.8).
#   bf16:   with random-init weights the sigma^2-amplified bf16 rounding (2^-9 per operand) saturates the clip on the high-sigma calls: per-call
#           PSNR(bf16, f32) is 8 - 20 dB above sigma 10 for the Type-I runs -- the bf16 mode does NOT reproduce the f32 guided call there, which
#           is why it is not the headline arithmetic.  Asserted: the floors it does hold, on the calls below sigma 1 (measured 46 - 102 dB)
#           and below sigma 0.03 (77 - 102 dB); everything else is printed and recorded.
BF16_TF_FLOOR_SIGMA_LT_1, BF16_TF_FLOOR_SIGMA_LT_003 = 40.0, 70.0


@pytest.mark.parametrize("opn,guid,cov,extra", E2E)
def test_e2e_teacher_forced(opn, guid, cov, extra):
    """End-to-end fidelity of the three arithmetic modes at full size (FFHQ architecture, 256 x 256, batch 2, random-init weights),
    TEACHER-FORCED: the exact-f32 mode runs a 20-step Heun `--ode` schedule once (39 guided calls from sigma 80 to 0.01) and every
    model call's input (x_i, sigma_i) is recorded; bf16x3 and bf16 are then evaluated on those SAME inputs and compared call by call
    with the teacher's outputs.  (A free-running comparison of two trajectories carries no information with random weights: the
    Type-I ODE is chaotic and two runs of one arithmetic land 20 - 40 dB apart -- DESIGN.md section 3; those numbers are only printed.)
      bf16x3: max-abs <= 1e-4 x max(2, sigma^2) on every call (2e-4, the bound of the full-size oracle comparisons above, up to sigma 1.4;
              the sigma^2 amplification of the guidance term above: 0.64 at sigma 80, measured 7.8e-2) AND per-call PSNR(bf16x3, f32) > 65 dB
              -- the PSNR floor is the effective guard at high sigma.  The guided output is discontinuous
              where |x0_raw| crosses 1 (VJP through clamp, condition.py:231), so calls whose clamp mask differs from the teacher's are
              re-evaluated with the teacher's mask imposed (stepwise path) after checking that the masks differ only at pixels within
              1e-4 x max(1, sigma / 5) of the boundary.
              The two non-chaotic runs (super-resolution Type-II, inpainting DPS) also assert the north_star tolerance end to end on the
              FREE-RUNNING trajectory: |PSNR_bf16x3 - PSNR_f32| < 1e-3 dB per image against the ground truth (measured 5e-7 / 2e-6 dB;
              both modes are bitwise reproducible).  The 100-step version of that assertion is test_e2e_100_steps_bf16x3_vs_f32_config2.
      bf16:   per-call PSNR(bf16, f32) floors, stated above.
    Recorded in gpurun_out/e2e_teacher_forced.jsonl."""
    import json, os
    import kdip_amd.unet as ku
    import kdip_amd.condition as kc
    import kdip_amd.sampling as ks
    from kdip_amd.evaluation import psnr
    B, STEPS = 2, 20
    m, sd, ocfg, hop, oop, meas, x0 = _setup("FFHQ", opn, "f32", B=B)
    D = ku.GaussianDiffusionTables()
    measd = (meas[0].cuda(), meas[1].cuda())
    xT = torch.randn(B, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) * 80
    sig = ks.get_sigmas_karras(STEPS, 0.01, 80, rho=7.0, device="cuda")

    def mk(model):
        return kc.ConditionOpenAIDenoiser(inner_model=model, diffusion=D, x0_cov_type=cov, recon_mse=None, operator=hop,
                                          measurement=measd, guidance=guid, zeta=extra.get("zeta"), device="cuda")
    teacher = _Teacher(mk(m))
    out_f32 = ks.sample_heun(teacher, xT.clone(), sig, disable=True).cpu()
    assert len(teacher.calls) == 2 * STEPS - 1 and torch.isfinite(out_f32).all()
    p_f32 = [float(psnr(out_f32[i:i + 1], x0[i:i + 1])) for i in range(B)]
    rec = {"operator": opn, "guidance": guid, "cov": cov, "steps": STEPS, "batch": B, "calls": len(teacher.calls), "psnr_f32_vs_gt": p_f32}
    checks = []
    free_dp = 0.0
    for dtype in ("bf16x3", "f16x3", "bf16"):      # (f16x3: the opt-in fp16-headed split, held to bf16x3's bounds; no call may need its bf16-headed redo)
        del m
        torch.cuda.empty_cache()
        m = ku.UNetModel(dtype=dtype, **ku.FFHQ_CONFIG); m.load_state_dict(sd)
        den = mk(m)
        flips_total, forced_calls, psnrs, errs, sigs = 0, 0, [], [], []
        for (x, s, out_t, raw_t) in teacher.calls:
            sv = ks._sigma_vec(x, s)
            out = den(x, sv)
            assert torch.isfinite(out).all()
            if dtype in ("bf16x3", "f16x3") and raw_t is not None:
                raw = den._stash[0]
                differ = (raw.abs() <= 1) != (raw_t.abs() <= 1)
                nd = int(differ.sum())
                if nd:
                    dist = float((raw_t[differ].abs() - 1).abs().max())
                    # only borderline pixels may flip.  x0_raw = a_t x c_in - b_t eps with b_t ~ sigma: an error of the UNet output (~5e-6 of its
                    # scale) moves x0_raw by ~sigma times that, so "borderline" is 1e-4 up to sigma 5 and 2e-5 sigma above (measured: 1.9e-4 at sigma 46)
                    assert nd <= 64 and dist < 1e-4 * max(1.0, 0.2 * s), (opn, s, nd, dist)
                    flips_total += nd
                    forced_calls += 1
                    out = _forced_mask_call(den, x, sv, raw_t)
            errs.append(float((out - out_t).abs().max()))
            psnrs.append(psnr_db(out, out_t))
            sigs.append(s)
        checks.append((dtype, sigs, errs, psnrs))
        if dtype in ("bf16x3", "f16x3"):
            print(f"\nteacher-forced {opn} {guid}/{cov} {dtype} ({getattr(m, 'x3_fallbacks', 0)} calls redone bf16-headed): {flips_total} borderline clamp pixels over {forced_calls} of {len(teacher.calls)} calls "
                  f"(re-evaluated with the teacher's mask); min per-call PSNR {min(psnrs):.1f} dB")
            print("  per call (sigma: max-abs, PSNR dB): " + "  ".join(f"{a:.3g}: {b:.1e}, {c:.0f}" for a, b, c in zip(sigs, errs, psnrs)))
        else:
            print(f"teacher-forced {opn} {guid}/{cov} bf16: per-call PSNR(bf16, f32) min {min(psnrs):.1f} / median {float(np.median(psnrs)):.1f} dB")
            print("  per call (sigma: max-abs, PSNR dB): " + "  ".join(f"{a:.3g}: {b:.1e}, {c:.0f}" for a, b, c in zip(sigs, errs, psnrs)))
        # free-running run of this mode from the same x_T: printed diagnostics only (never compared with a re-run)
        free = ks.sample_heun(den, xT.clone(), sig, disable=True).cpu()
        assert torch.isfinite(free).all()
        p = [float(psnr(free[i:i + 1], x0[i:i + 1])) for i in range(B)]
        dp = max(abs(u - v) for u, v in zip(p_f32, p))
        print(f"  free-running {dtype}: PSNR vs GT {p} (f32 {p_f32}), |dPSNR| {dp:.2e} dB, PSNR({dtype}, f32) {psnr_db(free, out_f32):.1f} dB  [diagnostic]")
        rec[dtype] = {"sigma": sigs, "call_max_abs": errs, "call_psnr_db": psnrs, "clamp_flips": flips_total, "forced_calls": forced_calls,
                      "free_running_psnr_vs_gt": p, "free_running_abs_dpsnr_db": dp}
        if dtype in ("bf16x3", "f16x3") and opn in ("super_resolution", "inpainting"):
            free_dp = max(dp, free_dp if dtype == "f16x3" else 0.0)      # asserted below (after the record is written)
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "e2e_teacher_forced.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n")
    if opn in ("super_resolution", "inpainting"):
        assert free_dp < 1e-3, (opn, free_dp)       # north_star: within 1e-3 dB PSNR end to end (20 Heun steps, free-running)
    for dtype, sigs, errs, psnrs in checks:
        for sg, e, pp in zip(sigs, errs, psnrs):
            if dtype in ("bf16x3", "f16x3"):
                assert e <= 1e-4 * max(2.0, sg * sg), (opn, dtype, sg, e)
                assert pp > 65.0, (opn, dtype, sg, pp)
            else:
                assert sg >= 1.0 or pp > BF16_TF_FLOOR_SIGMA_LT_1, (opn, dtype, sg, pp)
                assert sg >= 0.03 or pp > BF16_TF_FLOOR_SIGMA_LT_003, (opn, dtype, sg, pp)


def test_e2e_100_steps_bf16x3_vs_f32_config2():
    """BASELINE configs[2] at its full schedule length: FFHQ 256 x 256, 4x super-resolution, Type-II + PiGDM, 100 Heun steps (199 guided
    calls), batch 2 -- the split-precision mode against the exact-f32 mode from the same x_T.  This trajectory is reproducible (no VJP
    through the clamp), so the north_star tolerance applies as it stands: |PSNR_bf16x3 - PSNR_f32| < 1e-3 dB per image."""
    import kdip_amd.unet as ku
    import kdip_amd.condition as kc
    import kdip_amd.sampling as ks
    from kdip_amd.evaluation import psnr
    B = 2
    m, sd, ocfg, hop, oop, meas, x0 = _setup("FFHQ", "super_resolution", "f32", B=B)
    D = ku.GaussianDiffusionTables()
    measd = (meas[0].cuda(), meas[1].cuda())
    xT = torch.randn(B, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) * 80
    sig = ks.get_sigmas_karras(100, 0.01, 80, rho=7.0, device="cuda")
    outs = {}
    for dtype in ("f32", "bf16x3", "f16x3"):
        if dtype != "f32":
            del m
            torch.cuda.empty_cache()
            m = ku.UNetModel(dtype=dtype, **ku.FFHQ_CONFIG); m.load_state_dict(sd)
        den = kc.ConditionOpenAIDenoiser(inner_model=m, diffusion=D, x0_cov_type="pgdm", recon_mse=None, operator=hop, measurement=measd,
                                         guidance="II", device="cuda")
        outs[dtype] = ks.sample_heun(den, xT.clone(), sig, disable=True).cpu()
    pa = [float(psnr(outs["f32"][i:i + 1], x0[i:i + 1])) for i in range(B)]
    for dtype in ("bf16x3", "f16x3"):
        pc = [float(psnr(outs[dtype][i:i + 1], x0[i:i + 1])) for i in range(B)]
        dp = max(abs(u - v) for u, v in zip(pa, pc))
        print(f"\nconfigs[2] 100 Heun steps: PSNR vs GT f32 {pa} {dtype} {pc}  |dPSNR| {dp:.2e} dB  PSNR({dtype}, f32) {psnr_db(outs[dtype], outs['f32']):.1f} dB"
              + (f"  ({m.x3_fallbacks} of 199 calls redone bf16-headed)" if dtype == "f16x3" else ""))
        assert torch.isfinite(outs[dtype]).all() and dp < 1e-3, (dtype, dp)


CONFIGS = [
    # (id, model cfg, operator, guidance, cov, extra, v2/ortho, sampler, steps)
    ("cfg1_inpaint_dps_euler", "FFHQ", "inpainting", "dps", "dps", dict(zeta=1.0), None, "euler", 4),
    ("cfg2_gauss_typeI_convert_heun", "FFHQ", "gaussian_blur", "I", "convert", {}, None, "heun", 3),
    ("cfg3_sr4_typeII_pgdm", "FFHQ", "super_resolution", "II", "pgdm", {}, None, "heun", 3),
    ("cfg4_imagenet_motion_typeI_analytic", "IMAGENET", "motion_blur", "I", "analytic", {}, None, "heun", 3),
    ("cfg5_gauss_v2_dwt_autoI", "FFHQ", "gaussian_blur", "autoI", None, {}, "dwt", "heun", 3),
]


@pytest.mark.parametrize("cid,cfg,opn,guid,cov,extra,ortho,sampler,steps", CONFIGS)
def test_baseline_config_shapes_run(cid, cfg, opn, guid, cov, extra, ortho, sampler, steps):
    """Every BASELINE config's code path at full size, batch 4, bf16, a short schedule that ends at
    sigma_min: outputs finite, in range, and closer to the ground truth than the measurement's
    naive back-projection is not required (random weights) -- this is a does-it-run-at-size check."""
    import kdip_amd.unet as ku
    import kdip_amd.condition as kc
    import kdip_amd.external as ke
    import kdip_amd.sampling as ks
    m, sd, ocfg, hop, oop, meas, x0 = _setup(cfg, opn, "bf16", B=4, out_cov=ortho is not None)
    D = ku.GaussianDiffusionTables()
    measd = (meas[0].cuda(), meas[1].cuda())
    if ortho is None:
        rm = None
        if cov == "analytic":
            from helpers import synthetic_recon_mse
            rm = {k: v.cuda() for k, v in synthetic_recon_mse().items()}
        den = kc.ConditionOpenAIDenoiser(inner_model=m, diffusion=D, x0_cov_type=cov, recon_mse=rm, operator=hop,
                                         measurement=measd, guidance=guid, zeta=extra.get("zeta"), device="cuda")
    else:
        den = kc.ConditionOpenAIDenoiserV2(ke.OpenAIDenoiserV2(m, D, ortho_tf_type=ortho), operator=hop, measurement=measd,
                                           guidance=guid, mle_sigma_thres=1.0, device="cuda", ortho_tf_type=ortho)
    sig = ks.get_sigmas_karras(steps, 0.01, 80, rho=7.0, device="cuda")
    xT = torch.randn(4, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) * 80
    fn = ks.sample_heun if sampler == "heun" else ks.sample_euler
    out = fn(den, xT, sig, disable=True)
    assert out.shape == (4, 3, 256, 256) and torch.isfinite(out).all()
    assert float(out.abs().max()) < 5.0


@pytest.mark.parametrize("dt", ["f32", "bf16x3"])
def test_unet_vjp_properties_fullsize(dt):
    """Size-independent properties of the hand-written input-VJP at full size (FFHQ, f32 and bf16x3 modes, batch 2):
    linearity in the cotangent, and <c, J v> from a central finite difference of the forward == <J^T c, v> (checked over repeated runs: the forward has fp64-atomic-order noise ~1e-7)."""
    import kdip_amd.unet as ku
    m = ku.UNetModel(dtype=dt, **ku.FFHQ_CONFIG)
    m.load_state_dict(ku.synthetic_state_dict(seed=0, **ku.FFHQ_CONFIG))
    g = torch.Generator().manual_seed(9)
    x = (0.7 * smooth_image(2, 256, 4) + 0.3 * torch.randn(2, 3, 256, 256, generator=g)).cuda()
    t = torch.tensor([250.0, 40.0], device="cuda")
    c1 = torch.randn(2, 6, 256, 256, generator=g).cuda()
    c2 = torch.randn(2, 6, 256, 256, generator=g).cuda()
    m.forward(x, t)
    g1, g2, g12 = m.vjp(c1), m.vjp(c2), m.vjp(0.5 * c1 - 2.0 * c2)
    lin = float((g12 - (0.5 * g1 - 2.0 * g2)).abs().max() / g12.abs().max())
    assert lin < 2e-5, lin
    v = torch.randn(2, 3, 256, 256, generator=g).cuda()        # unit variance per element: the step must dominate fp32 noise
    eps = 5e-3
    fp = m.forward(x + eps * v, t).double()
    fm = m.forward(x - eps * v, t).double()
    lhs = ((fp - fm) / (2 * eps) * c1.double()).flatten(1).sum(1)          # <c, J v> per sample
    m.forward(x, t)
    rhs = (m.vjp(c1).double() * v.double()).flatten(1).sum(1)              # <J^T c, v>
    rel = float(((lhs - rhs).abs() / rhs.abs().clamp_min(1e-6)).max())
    print(f"\n{dt}: VJP linearity {lin:.1e}; directional derivative rel err {rel:.2e} (lhs {lhs.tolist()}, rhs {rhs.tolist()})")
    assert rel < 5e-3, (lhs, rhs)


def test_gn_coefficient_fold_bit_identical():
    """The large-map bf16 convs compute their GroupNorm staging coefficients from the statistics themselves (Conv3Fuse::fold_*: no
    gn_coef / gn_merge_stats / gn_bwd_coef launch between two convs).  Same arithmetic as the separate kernels: UNet forward and
    input-VJP at the FFHQ shape, batch 2, with the fold on and off, agree to the run-to-run noise of the fp64 statistics atomics
    (and the coefficients the VJP reads back are the ones the convs wrote)."""
    import kdip_amd._lib as L
    import kdip_amd.unet as ku
    lib = L.load()
    sd = ku.synthetic_state_dict(seed=0, **ku.FFHQ_CONFIG)
    m = ku.UNetModel(dtype="bf16", **ku.FFHQ_CONFIG)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 3, 256, 256, generator=g).cuda()
    t = torch.tensor([321.0, 321.0]).cuda()
    cot = torch.randn(2, 6, 256, 256, generator=g).cuda()
    res = {}
    try:
        for on in (1, 0, 1):
            L.check(lib.kdip_debug_gn_fold(on))
            out, _, _ = m.forward_raw(x, t, in_scale=0.7)
            res.setdefault(on, []).append((out.clone(), m.vjp(cot).clone()))
    finally:
        L.check(lib.kdip_debug_gn_fold(1))
    noise_o = float((res[1][0][0] - res[1][1][0]).abs().max())          # two runs of the SAME mode: atomics-order noise
    noise_g = float((res[1][0][1] - res[1][1][1]).abs().max())
    d_o = float((res[1][0][0] - res[0][0][0]).abs().max())
    d_g = float((res[1][0][1] - res[0][0][1]).abs().max())
    so, sg = float(res[0][0][0].abs().max()), float(res[0][0][1].abs().max())
    print(f"\nfold on vs off: out {d_o:.3e} (same-mode noise {noise_o:.3e}, scale {so:.2f}); vjp {d_g:.3e} (noise {noise_g:.3e}, scale {sg:.2f})")
    assert d_o <= max(4 * noise_o, 2e-2 * so) and d_g <= max(4 * noise_g, 2e-2 * sg)
