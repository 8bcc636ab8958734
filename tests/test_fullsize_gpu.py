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


@pytest.mark.parametrize("sigma_v", [1.5, 0.12])
def test_ffhq_type1_convert_fullsize(sigma_v):
    """BASELINE configs[1] shape at batch 1: f32 mode within 2e-3 max-abs of the oracle (incl. CG
    branch at sigma 0.12); bf16 mode reported as PSNR."""
    import kdip_amd.unet as ku
    import kdip_amd.condition as kc
    from oracle import condition as ocond
    m, sd, ocfg, hop, oop, meas, x0 = _setup("FFHQ", "gaussian_blur", "f32")
    x = x0 + sigma_v * torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(11))
    ref = ocond.GuidedDenoiser(sd, ocfg, oop, meas, "I", x0_cov_type="convert")(x, torch.tensor([sigma_v]))
    D = ku.GaussianDiffusionTables()
    hm = kc.ConditionOpenAIDenoiser(inner_model=m, diffusion=D, x0_cov_type="convert", recon_mse=None, operator=hop,
                                    measurement=(meas[0].cuda(), meas[1].cuda()), guidance="I", device="cuda")
    hat = hm(x.cuda(), torch.tensor([sigma_v], device="cuda")).cpu()
    err = float((hat - ref).abs().max())
    assert err < 2e-3, err
    del m, hm
    m2 = ku.UNetModel(dtype="bf16", **ku.FFHQ_CONFIG); m2.load_state_dict(sd)
    hm2 = kc.ConditionOpenAIDenoiser(inner_model=m2, diffusion=D, x0_cov_type="convert", recon_mse=None, operator=hop,
                                     measurement=(meas[0].cuda(), meas[1].cuda()), guidance="I", device="cuda")
    hat2 = hm2(x.cuda(), torch.tensor([sigma_v], device="cuda")).cpu()
    p = psnr_db(hat2, ref)
    print(f"\nFFHQ full-size sigma={sigma_v}: f32 max-abs {err:.2e}; bf16 PSNR(hip, oracle) {p:.1f} dB")
    assert p > 30.0


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
    e1 = float((out.cpu() - out_ref.detach()).abs().max() / out_ref.abs().max())
    e2 = float((vjp.cpu() - vjp_ref).abs().max() / vjp_ref.abs().max())
    print(f"\nImageNet-256 UNet f32: fwd rel err {e1:.2e}, vjp rel err {e2:.2e}")
    assert e1 < 5e-4 and e2 < 5e-4


CONFIGS = [
    # (id, model cfg, operator, guidance, cov, extra, v2/ortho, sampler, steps)
    ("cfg1_inpaint_dps_euler", "FFHQ", "inpainting", "dps", "dps", dict(zeta=1.0), None, "euler", 4),
    ("cfg2_gauss_typeI_convert_heun", "FFHQ", "gaussian_blur", "I", "convert", {}, None, "heun", 3),
    ("cfg3_sr4_typeII_pgdm", "FFHQ", "super_resolution", "II", "pgdm", {}, None, "heun", 3),
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
        den = kc.ConditionOpenAIDenoiser(inner_model=m, diffusion=D, x0_cov_type=cov, recon_mse=None, operator=hop,
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


def test_unet_vjp_properties_fullsize():
    """Size-independent properties of the hand-written input-VJP at full size (FFHQ, f32 mode, batch 2):
    linearity in the cotangent, and <c, J v> from a central finite difference of the forward == <J^T c, v> (checked over repeated runs: the forward has fp64-atomic-order noise ~1e-7)."""
    import kdip_amd.unet as ku
    m = ku.UNetModel(dtype="f32", **ku.FFHQ_CONFIG)
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
    print(f"\nVJP linearity {lin:.1e}; directional derivative rel err {rel:.2e} (lhs {lhs.tolist()}, rhs {rhs.tolist()})")
    assert rel < 5e-3, (lhs, rhs)
