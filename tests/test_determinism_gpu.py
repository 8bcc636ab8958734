"""GPU: the parity modes (f32, bf16x3) are bit-reproducible -- every cross-block reduction (GroupNorm statistics, fused conv
statistics, split-K partial sums, CG dot products) is made in a fixed order (csrc/det.h), as the reference's CPU path is
deterministic (guided_diffusion/unet.py:182-213, nn.py:17-19).  Two runs of one UNet call / guided call / 20-step sampler run
must be BITWISE equal; the fixed-order sums must agree with the atomics they replace to fp32 rounding."""
import numpy as np
import pytest
import torch

from helpers import smooth_image

pytestmark = pytest.mark.gpu


def _ffhq(dt):
    import kdip_amd.unet as ku
    m = ku.UNetModel(dtype=dt, **ku.FFHQ_CONFIG)
    m.load_state_dict(ku.synthetic_state_dict(seed=0, **ku.FFHQ_CONFIG))
    return m


@pytest.mark.parametrize("dt", ["f32", "bf16x3", "f16x3"])
def test_unet_forward_vjp_bitwise_reproducible(dt):
    """FFHQ architecture at 256 x 256, batch 3 (not a power of two: ragged tile / chunk counts): forward + input-VJP three times ->
    identical bits.  With the fixed-order reductions switched off (the bf16 mode's atomics) the same call agrees to fp32 rounding,
    i.e. the ordered sums compute the same statistics."""
    m = _ffhq(dt)
    g = torch.Generator().manual_seed(9)
    x = (0.7 * smooth_image(3, 256, 4) + 0.3 * torch.randn(3, 3, 256, 256, generator=g)).cuda()
    t = torch.tensor([250.0, 40.0, 700.0], device="cuda")
    cot = torch.randn(3, 6, 256, 256, generator=g).cuda()
    runs = []
    for _ in range(3):
        out, _, _ = m.forward_raw(x, t, in_scale=0.8)
        runs.append((out.clone(), m.vjp(cot).clone()))
    for o, v in runs[1:]:
        assert torch.equal(o, runs[0][0]) and torch.equal(v, runs[0][1])
    assert torch.isfinite(runs[0][0]).all() and torch.isfinite(runs[0][1]).all()
    assert m.set_deterministic(False) is True
    try:
        out, _, _ = m.forward_raw(x, t, in_scale=0.8)
        v = m.vjp(cot)
    finally:
        m.set_deterministic(True)
    eo = float((out - runs[0][0]).abs().max() / runs[0][0].abs().max())
    ev = float((v - runs[0][1]).abs().max() / runs[0][1].abs().max())
    print(f"\n{dt}: three runs bitwise equal; fixed-order vs atomics: forward rel {eo:.2e}, vjp rel {ev:.2e}")
    assert eo < 2e-5 and ev < 5e-5
    out2, _, _ = m.forward_raw(x, t, in_scale=0.8)          # back in the deterministic mode (re-planned workspace): the same bits again
    assert torch.equal(out2, runs[0][0]) and torch.equal(m.vjp(cot), runs[0][1])


@pytest.mark.parametrize("dt", ["f32", "bf16x3", "f16x3"])
def test_tiny_unet_bitwise_reproducible(dt):
    """The tiny golden-vector architecture (32 base channels: one channel per GroupNorm group, the general path of the ordered
    statistics; attention at 32 x 32), batch 5."""
    import kdip_amd.unet as ku
    cfg = dict(image_size=64, model_channels=32, num_res_blocks=1, attention_resolutions="32", channel_mult=(1, 2))
    m = ku.UNetModel(dtype=dt, **cfg)
    m.load_state_dict(ku.synthetic_state_dict(seed=1, **cfg))
    g = torch.Generator().manual_seed(2)
    x = torch.randn(5, 3, 64, 64, generator=g).cuda()
    t = torch.tensor([3.0, 999.0, 500.5, 20.0, 77.0], device="cuda")
    cot = torch.randn(5, 6, 64, 64, generator=g).cuda()
    o0, _, _ = m.forward_raw(x, t)
    v0 = m.vjp(cot)
    o0, v0 = o0.clone(), v0.clone()
    for _ in range(2):
        o, _, _ = m.forward_raw(x, t)
        assert torch.equal(o, o0) and torch.equal(m.vjp(cot), v0)


RUNS = [("gaussian_blur", "I", "convert", {}), ("inpainting", "dps", "dps", dict(zeta=1.0)), ("super_resolution", "II", "pgdm", {})]


@pytest.mark.parametrize("dt", ["f32", "bf16x3", "f16x3"])
@pytest.mark.parametrize("opn,guid,cov,extra", RUNS)
def test_sampler_run_bitwise_reproducible(dt, opn, guid, cov, extra):
    """A 20-step Heun `--ode` run (39 guided calls, sigma 80 -> 0.01: closed-form and CG branches of the mat-solver, hand-written VJP,
    clamp-gradient mask) at 256 x 256, batch 2, twice from the same x_T: the two results are bitwise equal -- the chaotic random-weight
    trajectory included (two f32 runs used to land 20 - 40 dB apart)."""
    import kdip_amd.unet as ku
    import kdip_amd.condition as kc
    import kdip_amd.sampling as ks
    from test_fullsize_gpu import _setup
    B = 2
    m, sd, ocfg, hop, oop, meas, x0 = _setup("FFHQ", opn, dt, B=B)
    D = ku.GaussianDiffusionTables()
    den = kc.ConditionOpenAIDenoiser(inner_model=m, diffusion=D, x0_cov_type=cov, recon_mse=None, operator=hop,
                                     measurement=(meas[0].cuda(), meas[1].cuda()), guidance=guid, zeta=extra.get("zeta"), device="cuda")
    xT = torch.randn(B, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) * 80
    sig = ks.get_sigmas_karras(20, 0.01, 80, rho=7.0, device="cuda")
    a = ks.sample_heun(den, xT.clone(), sig, disable=True).clone()
    b = ks.sample_heun(den, xT.clone(), sig, disable=True)
    assert torch.isfinite(a).all()
    nd = int((a != b).sum())
    assert nd == 0, (dt, opn, nd, float((a - b).abs().max()))


@pytest.mark.parametrize("dt", ["f32", "bf16x3", "f16x3"])
@pytest.mark.parametrize("cid", ["cfg3_imagenet_motion_typeI_analytic", "cfg4_gauss_v2_dwt_autoI"])
def test_other_baseline_configs_bitwise_reproducible(dt, cid):
    """BASELINE configs[3] (ImageNet-256 architecture: 16 attention blocks on the fp32 GEMM + softmax path, analytic covariance) and
    configs[4] (V2 denoiser, DWT-Var covariance, auto Type-I: CG with the Haar DWT in the matvec below sigma 1): a 4-step Heun run
    ending at sigma_min, batch 2, twice -> identical bits."""
    import kdip_amd.unet as ku
    import kdip_amd.condition as kc
    import kdip_amd.external as ke
    import kdip_amd.sampling as ks
    from test_fullsize_gpu import _setup
    from helpers import synthetic_recon_mse
    B = 2
    D = ku.GaussianDiffusionTables()
    if cid.startswith("cfg3"):
        m, sd, ocfg, hop, oop, meas, x0 = _setup("IMAGENET", "motion_blur", dt, B=B)
        rm = {k: v.cuda() for k, v in synthetic_recon_mse().items()}
        den = kc.ConditionOpenAIDenoiser(inner_model=m, diffusion=D, x0_cov_type="analytic", recon_mse=rm, operator=hop,
                                         measurement=(meas[0].cuda(), meas[1].cuda()), guidance="I", device="cuda")
    else:
        m, sd, ocfg, hop, oop, meas, x0 = _setup("FFHQ", "gaussian_blur", dt, B=B, out_cov=True)
        den = kc.ConditionOpenAIDenoiserV2(ke.OpenAIDenoiserV2(m, D, ortho_tf_type="dwt"), operator=hop, measurement=(meas[0].cuda(), meas[1].cuda()),
                                           guidance="autoI", mle_sigma_thres=1.0, device="cuda", ortho_tf_type="dwt")
    xT = torch.randn(B, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) * 80
    sig = ks.get_sigmas_karras(4, 0.01, 80, rho=7.0, device="cuda")
    a = ks.sample_heun(den, xT.clone(), sig, disable=True).clone()
    b = ks.sample_heun(den, xT.clone(), sig, disable=True)
    assert torch.isfinite(a).all()
    nd = int((a != b).sum())
    assert nd == 0, (dt, cid, nd, float((a - b).abs().max()))


@pytest.mark.parametrize("dtype", ["f32", "f16x3"])
def test_deferred_statistics_finish_bit_identical(dtype):
    """Deterministic modes: a forward conv with fused GroupNorm statistics leaves their fixed-order finish pass to the consuming GroupNorm, which runs it inside
    its coefficient kernel (conv_stats_finish_coef: one launch less in the chain between two convs).  Same summation tree, same coefficient arithmetic:
    UNet forward and input-VJP at the FFHQ shape are BITWISE those of the finish-behind-the-conv path (kdip_debug_defer_finish)."""
    import kdip_amd._lib as L
    import kdip_amd.unet as ku
    lib = L.load()
    sd = ku.synthetic_state_dict(seed=0, **ku.FFHQ_CONFIG)
    m = ku.UNetModel(dtype=dtype, **ku.FFHQ_CONFIG)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 3, 256, 256, generator=g).cuda()
    t = torch.tensor([321.0, 47.0]).cuda()
    cot = torch.randn(2, 6, 256, 256, generator=g).cuda()
    res = {}
    try:
        for on in (1, 0):
            L.check(lib.kdip_debug_defer_finish(on))
            out, _, _ = m.forward_raw(x, t, in_scale=0.7)
            res[on] = (out.clone(), m.vjp(cot).clone())
    finally:
        L.check(lib.kdip_debug_defer_finish(1))
    assert torch.isfinite(res[1][0]).all() and torch.isfinite(res[1][1]).all()
    assert torch.equal(res[1][0], res[0][0]) and torch.equal(res[1][1], res[0][1])
