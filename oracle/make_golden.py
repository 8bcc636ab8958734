"""DEV-ONLY: validate the oracle against the real reference and emit tests/golden/*.npz.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference);
the committed fixtures are what travels.  Usage:  python -m oracle.make_golden

For every fixture the reference output is (1) compared with the oracle
restatement here (assert) and (2) saved as inputs + expected outputs.
"""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import refimport                                     # noqa: E402
from oracle import tables as otab, unet as ounet, operators as oops   # noqa: E402
from oracle import transforms as otf, solvers as osol, condition as ocond, sampling as osamp  # noqa: E402
from oracle import analytic as oana  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def smooth_image(B, size, seed=1):
    """Seeded smooth field in [-1,1]."""
    g = torch.Generator().manual_seed(seed)
    r = torch.rand(B, 3, size, size, generator=g) * 2 - 1
    rp = torch.nn.functional.pad(r, (4, 4, 4, 4), mode="circular")
    return (3 * torch.nn.functional.avg_pool2d(rp, 9, 1)).clamp(-1, 1)


def maxdiff(a, b):
    return float((a.detach() - b.detach()).abs().max())


def check(name, ref, ora, tol):
    d = maxdiff(ref, ora)
    print(f"  {name:55s} max|ref-oracle| = {d:.3e}  (tol {tol:g})")
    assert d <= tol, name
    return d


def build_ref_model(ns, cfg_dict, sd, out_cov=False):
    su, um = ns.su, ns.um
    mc = dict(image_size=cfg_dict["image_size"], num_channels=cfg_dict["model_channels"],
              num_res_blocks=cfg_dict["num_res_blocks"],
              attention_resolutions=cfg_dict["attention_resolutions"])
    if "channel_mult" in cfg_dict:
        mc["channel_mult"] = ",".join(str(c) for c in cfg_dict["channel_mult"])
    args = um.create_argparser(mc).parse_args([])
    keys = su.model_and_diffusion_defaults().keys()
    model, diffusion = su.create_model_and_diffusion(**{k: getattr(args, k) for k in keys})
    unet_sd = {k: v for k, v in sd.items() if not k.startswith("out_cov.")}
    model.load_state_dict(unet_sd, strict=True)
    return model.eval(), diffusion


def main():
    os.makedirs(GOLD, exist_ok=True)
    ns = refimport.import_reference()
    cc, cm, cu, ks, ke = ns.cc, ns.cm, ns.cu, ns.ks, ns.ke

    # ---------------------------------------------------------------- tables ----
    print("[tables]")
    D = otab.DiffusionTables()
    sig_ref = ks.get_sigmas_karras(100, 0.01, 80, rho=7.0)
    check("get_sigmas_karras(100)", sig_ref, otab.get_sigmas_karras(100, 0.01, 80), 0)
    sig20 = ks.get_sigmas_karras(20, 0.01, 80, rho=7.0)
    cfg = ounet.UNetConfig(**ounet.TINY)
    sd = ounet.init_state_dict(cfg, seed=0)
    model, diffusion = build_ref_model(ns, ounet.TINY, sd)
    den = ke.OpenAIDenoiser(model, diffusion)
    probe = torch.cat([sig_ref[:-1], torch.tensor([80.0, 10.0, 1.0, 0.2, 0.1999, 0.05, 0.01, 157.0, 0.005, 3.3])])
    t_ref = den.sigma_to_t(probe)
    check("sigma_to_t(frac)", t_ref, D.sigma_to_t(probe), 0)
    for nm in ["alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
               "posterior_mean_coef1", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"]:
        assert np.array_equal(getattr(diffusion, nm), getattr(D, nm)), nm
    assert np.array_equal(np.log(diffusion.betas), D.log_betas)
    temb_t = torch.tensor([0.0, 10.0, 57.0, 258.5, 673.0, 929.0])
    from guided_diffusion.nn import timestep_embedding as ref_temb
    check("timestep_embedding", ref_temb(temb_t, 32), otab.timestep_embedding(temb_t, 32), 0)
    np.savez_compressed(os.path.join(GOLD, "tables.npz"),
                        sigmas100=sig_ref.numpy(), sigmas20=sig20.numpy(), probe=probe.numpy(),
                        t_frac=t_ref.numpy(), t_floor=t_ref.long().numpy(),
                        temb_t=temb_t.numpy(), temb=ref_temb(temb_t, 32).numpy(),
                        alphas_cumprod=diffusion.alphas_cumprod)

    # ------------------------------------------------------------------ unet ----
    print("[unet tiny]")
    g = torch.Generator().manual_seed(5)
    xu = torch.randn(2, 3, 64, 64, generator=g)
    tu = torch.tensor([57.0, 673.25])
    with torch.no_grad():
        o_ref, h_ref = model(xu, tu, return_feature=True)
        o_ora, h_ora = ounet.unet_forward(sd, cfg, xu, tu, return_feature=True)
    check("UNetModel.forward out", o_ref, o_ora, 2e-5)
    check("UNetModel.forward feature", h_ref, h_ora, 2e-5)
    # VJP (cotangent on eps channels)
    cot = torch.randn(2, 6, 64, 64, generator=g)
    xr = xu.clone().requires_grad_()
    g_ref = torch.autograd.grad((model(xr, tu) * cot).sum(), xr)[0]
    xo = xu.clone().requires_grad_()
    g_ora = torch.autograd.grad((ounet.unet_forward(sd, cfg, xo, tu) * cot).sum(), xo)[0]
    check("UNet input-VJP", g_ref, g_ora, 2e-5)
    np.savez_compressed(os.path.join(GOLD, "unet_tiny.npz"), x=xu.numpy(), t=tu.numpy(),
                        out=o_ref.numpy(), feature=h_ref.numpy(), cot=cot.numpy(), vjp=g_ref.numpy())

    # ------------------------------------------------------------- operators ----
    print("[operators 64x64]")
    S = 64
    x0 = smooth_image(1, S, seed=1)
    op_cfgs = {
        "gaussian_blur": dict(in_shape=(1, 3, S, S), kernel_size=61, intensity=3.0, sigma_s=0.05),
        "motion_blur": dict(in_shape=(1, 3, S, S), kernel_size=61, intensity=0.5, sigma_s=0.05),
        "super_resolution": dict(in_shape=(1, 3, S, S), scale_factor=4, sigma_s=0.05),
        "inpainting": dict(sigma_s=0.05, mask_opt=dict(mask_type="random", mask_prob_range=(0.5, 0.5), image_size=S)),
    }
    ref_ops, ora_ops, meas_ref, meas_ora = {}, {}, {}, {}
    opdump = {"x0": x0.numpy()}
    for name, kw in op_cfgs.items():
        with refimport.reference_cwd():
            np.random.seed(0)
            rop = cm.get_operator(name, device="cpu", **kw)
        np.random.seed(0)
        oop = oops.get_operator(name, **kw)
        torch.manual_seed(2)
        y_r, yf_r = rop.forward(x0.clone(), flatten=True)
        torch.manual_seed(2)
        y_o, yf_o = oop.forward(x0.clone(), flatten=True)
        check(f"{name}.forward y", y_r, y_o, 1e-6)
        check(f"{name}.forward y_flat", yf_r, yf_o, 1e-6)
        y_nl = rop.forward(x0.clone(), noiseless=True)
        check(f"{name}.forward noiseless", y_nl, oop.forward(x0.clone(), noiseless=True), 1e-6)
        # restore pre_calculated for the noisy y (forward refreshes it)
        torch.manual_seed(2); rop.forward(x0.clone(), flatten=True)
        torch.manual_seed(2); oop.forward(x0.clone(), flatten=True)
        at_r = rop.transpose(yf_r, flatten=True)
        check(f"{name}.transpose(flat)", at_r, oop.transpose(yf_o, flatten=True), 1e-6)
        opdump.update({f"{name}.y": y_r.numpy(), f"{name}.y_flat": yf_r.numpy(),
                       f"{name}.y_noiseless": y_nl.numpy(), f"{name}.ATy": at_r.numpy()})
        if name == "inpainting":
            assert torch.equal(rop.mask, oop.mask)
            opdump["inpainting.mask_bits"] = np.packbits(rop.mask[0, 0].numpy().astype(np.uint8))
        else:
            FB = rop.pre_calculated[0]
            check(f"{name} FB", torch.view_as_real(FB), torch.view_as_real(oop.pre_calculated[0]), 1e-6)
            opdump[f"{name}.FB"] = FB.numpy()
        ref_ops[name], ora_ops[name] = rop, oop
        meas_ref[name], meas_ora[name] = (y_r, yf_r), (y_o, yf_o)
    # full-size mask for numpy seed 0 (bit-exact contract)
    with refimport.reference_cwd():
        np.random.seed(0)
        rop256 = cm.get_operator("inpainting", device="cpu", sigma_s=0.05,
                                 mask_opt=dict(mask_type="random", mask_prob_range=(0.5, 0.5), image_size=256))
    np.random.seed(0)
    m256 = oops.random_mask(256, (0.5, 0.5))
    assert torch.equal(rop256.mask, m256)
    opdump["inpainting.mask256_bits"] = np.packbits(m256[0, 0].numpy().astype(np.uint8))
    idx = torch.where(m256 > 0)
    opdump["inpainting.mask256_first_idx"] = torch.stack(idx[-3:])[:, :64].numpy()
    # full-size SR resizer measurement on a 256 image (the 16-tap antialiased cubic)
    x256 = smooth_image(1, 256, seed=1)
    with refimport.reference_cwd():
        rsr = cm.get_operator("super_resolution", device="cpu", in_shape=(1, 3, 256, 256), scale_factor=4, sigma_s=0.05)
    osr = oops.get_operator("super_resolution", in_shape=(1, 3, 256, 256), scale_factor=4, sigma_s=0.05)
    ysr = rsr.forward(x256.clone(), noiseless=True)
    check("super_resolution 256 resizer", ysr, osr.forward(x256.clone(), noiseless=True), 1e-6)
    opdump["sr256.y_noiseless"] = ysr.numpy()
    w16, fov16 = oops.resizer_contributions(256, 64, 0.25)
    assert torch.equal(rsr.down_sample.field_of_view[0], torch.tensor(fov16.T)), "resizer fov"
    np.savez_compressed(os.path.join(GOLD, "operators.npz"), **opdump)

    # ------------------------------------------------------------ transforms ----
    print("[transforms]")
    xt = torch.randn(1, 3, 64, 64, generator=g)
    dct_ref = cu.DiscreteCosineTransform()
    c_ref = dct_ref.forward(xt)
    check("dct", c_ref, otf.dct_ortho(xt), 5e-6)
    check("idct", dct_ref.transpose(c_ref), otf.idct_ortho(c_ref), 5e-6)
    w = otf.dwt_haar(xt)
    check("haar perfect reconstruction", xt, otf.idwt_haar(w), 1e-5)
    assert abs(float(w.norm() / xt.norm()) - 1) < 1e-6
    np.savez_compressed(os.path.join(GOLD, "transforms.npz"), x=xt.numpy(), dct=c_ref.numpy())

    # --------------------------------------------------------- guided calls ----
    print("[guided calls, tiny UNet, 64x64]")
    gd_dump = {}
    recon_mse = {"sigmas": otab.get_sigmas_karras(1000, 0.01, 80)[:-1],
                 "mse_list": (lambda s: s ** 2 / (1 + s ** 2) * 0.5)(otab.get_sigmas_karras(1000, 0.01, 80)[:-1])}
    modes = [("I", "convert", {}), ("II", "convert", {}), ("II", "pgdm", {}), ("dps", "dps", dict(zeta=1.0)),
             ("pgdm", "pgdm", {}), ("I", "analytic", {}), ("diffpir", "diffpir", dict(lambda_=7.0)),
             ("uncond", "convert", {}), ("dps+mle", "convert", dict(zeta=1.0))]
    for name in op_cfgs:
        for guidance, cov, extra in modes:
            for sigma_v in (1.5, 0.12):
                gx = torch.Generator().manual_seed(11)
                x = x0 + sigma_v * torch.randn(1, 3, S, S, generator=gx)
                sigma = torch.tensor([sigma_v])
                rmodel = cc.ConditionOpenAIDenoiser(
                    inner_model=model, diffusion=diffusion, x0_cov_type=cov,
                    recon_mse={k: v.clone() for k, v in recon_mse.items()},
                    operator=ref_ops[name], measurement=meas_ref[name], guidance=guidance,
                    zeta=extra.get("zeta"), lambda_=extra.get("lambda_"), mle_sigma_thres=0.2, device="cpu").eval()
                omodel = ocond.GuidedDenoiser(sd, cfg, ora_ops[name], meas_ora[name], guidance, x0_cov_type=cov,
                                              recon_mse=recon_mse, zeta=extra.get("zeta"),
                                              lambda_=extra.get("lambda_"), mle_sigma_thres=0.2)
                h_r = rmodel(x.clone(), sigma)
                h_o = omodel(x.clone(), sigma)
                key = f"{name}|{guidance}|{cov}|{sigma_v}"
                check(key, h_r, h_o, 2e-4)
                gd_dump[key] = h_r.numpy()
    gd_dump["x0"] = x0.numpy()
    np.savez_compressed(os.path.join(GOLD, "guided_calls.npz"), **gd_dump)

    # ------------------------------------------- tmpd covariance, STSL guidance ----
    print("[tmpd / stsl, tiny UNet]")
    ex_dump = {}
    ex_modes = [("I", "tmpd", {}), ("II", "tmpd", {}), ("stsl", "dps", dict(zeta=1.0, eta=0.5, num_hutchinson_samples=2)),
                ("stsl+mle", "convert", dict(zeta=1.0, eta=0.5, num_hutchinson_samples=2))]
    for name in ("gaussian_blur", "inpainting"):
        for guidance, cov, extra in ex_modes:
            if cov == "tmpd" and name == "inpainting":
                # with random weights the TMPD "variance" sigma^2 J^T 1 goes negative; the inpainting system
                # sigma_s^2 + m*C is then indefinite and CG (scipy or any) is ill-defined -- not a parity case
                continue
            for sigma_v in (1.5, 0.12):
                gx = torch.Generator().manual_seed(11)
                x = x0 + sigma_v * torch.randn(1, 3, S, S, generator=gx)
                sigma = torch.tensor([sigma_v])
                rmodel = cc.ConditionOpenAIDenoiser(
                    inner_model=model, diffusion=diffusion, x0_cov_type=cov, recon_mse=None,
                    operator=ref_ops[name], measurement=meas_ref[name], guidance=guidance,
                    zeta=extra.get("zeta"), eta=extra.get("eta"), num_hutchinson_samples=extra.get("num_hutchinson_samples"),
                    mle_sigma_thres=0.2, device="cpu").eval()
                omodel = ocond.GuidedDenoiser(sd, cfg, ora_ops[name], meas_ora[name], guidance, x0_cov_type=cov,
                                              zeta=extra.get("zeta"), eta=extra.get("eta"),
                                              num_hutchinson_samples=extra.get("num_hutchinson_samples"), mle_sigma_thres=0.2)
                torch.manual_seed(5)
                h_r = rmodel(x.clone(), sigma)
                torch.manual_seed(5)
                h_o = omodel(x.clone(), sigma)
                key = f"{name}|{guidance}|{cov}|{sigma_v}"
                check(key, h_r, h_o, 5e-4)
                ex_dump[key] = h_r.numpy()
    np.savez_compressed(os.path.join(GOLD, "guided_calls_extra.npz"), **ex_dump)

    # ------------------------------------------------- analytic-variance estimator ----
    print("[analytic variance (OpenAIDenoiser + Monte-Carlo MSE)]")
    sig_av = ks.get_sigmas_karras(5, 0.01, 80, rho=7.0)
    batches = [smooth_image(2, S, seed=21), smooth_image(2, S, seed=22)]
    torch.manual_seed(9)
    mse_ref = []
    with torch.no_grad():
        for sgm in sig_av:
            m_ = 0
            for xb in batches:
                hat = den(xb + torch.randn_like(xb) * sgm, sgm.repeat(2))
                m_ = m_ + (xb - hat).pow(2).mean()
            mse_ref.append(m_ / len(batches))
    mse_ref = torch.stack(mse_ref)
    torch.manual_seed(9)
    est = oana.estimate_recon_mse(sd, cfg, batches, sig_av)
    check("recon_mse list", mse_ref, est["mse_list"], 1e-5)
    np.savez_compressed(os.path.join(GOLD, "analytic_variance.npz"), sigmas=sig_av.numpy(), mse_list=mse_ref.numpy())

    # ------------------------------------------------------- V2 (out_cov head) ----
    print("[V2 calls, ortho_tf=None]")
    sd2 = ounet.init_state_dict(cfg, seed=0, out_cov=True)
    # TINY has 32 feature channels; the reference hard-codes Conv2d(128, 6, 1) (external.py:141)
    den2 = ke.OpenAIDenoiserV2(model, diffusion)
    den2.out_cov = torch.nn.Conv2d(32, 6, 1)
    den2.out_cov.weight.data.copy_(sd2["out_cov.weight"]); den2.out_cov.bias.data.copy_(sd2["out_cov.bias"])
    v2_dump = {}
    for name in ("gaussian_blur", "inpainting", "super_resolution"):
        for guidance in ("I", "II"):
            for sigma_v in (1.5, 0.12):
                gx = torch.Generator().manual_seed(11)
                x = x0 + sigma_v * torch.randn(1, 3, S, S, generator=gx)
                sigma = torch.tensor([sigma_v])
                rmodel = cc.ConditionOpenAIDenoiserV2(den2, operator=ref_ops[name], measurement=meas_ref[name],
                                                      guidance=guidance, mle_sigma_thres=1.0, device="cpu").eval()
                omodel = ocond.GuidedDenoiser(sd2, cfg, ora_ops[name], meas_ora[name], guidance,
                                              mle_sigma_thres=1.0, v2=True)
                h_r = rmodel(x.clone(), sigma)
                h_o = omodel(x.clone(), sigma)
                key = f"{name}|{guidance}|v2|{sigma_v}"
                check(key, h_r, h_o, 2e-4)
                v2_dump[key] = h_r.numpy()
    np.savez_compressed(os.path.join(GOLD, "guided_calls_v2.npz"), **v2_dump)

    # -------------------------------------------------------------- sampler ----
    print("[sampler trajectories, 4 steps, --ode]")
    sig4 = ks.get_sigmas_karras(4, 0.01, 80, rho=7.0)
    sm_dump = {"sigmas": sig4.numpy()}
    name = "gaussian_blur"
    for sampler in ("heun", "euler"):
        rmodel = cc.ConditionOpenAIDenoiser(inner_model=model, diffusion=diffusion, x0_cov_type="convert",
                                            recon_mse=None, operator=ref_ops[name], measurement=meas_ref[name],
                                            guidance="I", mle_sigma_thres=0.2, device="cpu").eval()
        omodel = ocond.GuidedDenoiser(sd, cfg, ora_ops[name], meas_ora[name], "I", x0_cov_type="convert")
        xT = torch.randn(1, 3, S, S, generator=torch.Generator().manual_seed(3)) * 80
        rf = ks.sample_heun if sampler == "heun" else ks.sample_euler
        of = osamp.sample_heun if sampler == "heun" else osamp.sample_euler
        x_r = rf(rmodel, xT.clone(), sig4, disable=True)
        x_o = of(omodel, xT.clone(), sig4)
        check(f"sample_{sampler} 4 steps ode", x_r, x_o, 5e-4)
        sm_dump[f"{sampler}.x0"] = x_r.detach().numpy()
        # churn: same global RNG stream
        torch.manual_seed(7)
        x_rc = rf(rmodel, xT.clone(), sig4, disable=True, s_churn=80, s_tmin=0.05, s_tmax=50, s_noise=1.003)
        torch.manual_seed(7)
        x_oc = of(omodel, xT.clone(), sig4, s_churn=80, s_tmin=0.05, s_tmax=50, s_noise=1.003)
        check(f"sample_{sampler} 4 steps churn", x_rc, x_oc, 5e-4)
        sm_dump[f"{sampler}.x0_churn"] = x_rc.detach().numpy()
    sm_dump["xT"] = xT.numpy()
    np.savez_compressed(os.path.join(GOLD, "sampler.npz"), **sm_dump)
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    main()
