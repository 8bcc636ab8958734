"""DEV-ONLY: later additions to the fixtures: sample_dpmpp_2m trajectory (k_diffusion/sampling.py:583-605) and box / extreme
inpainting masks (condition/measurements.py:264-320), from the real reference.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference).  Usage: python -m oracle.make_golden_extra
Same tiny model / Gaussian-blur operator / seeds as oracle.make_golden's sampler section; writes
tests/golden/sampler_dpmpp2m.npz and tests/golden/masks_box.npz (inputs + expected outputs) after asserting reference == oracle.
"""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refimport, unet as ounet, operators as oops, condition as ocond, sampling as osamp   # noqa: E402
from oracle.make_golden import build_ref_model, smooth_image, check, GOLD                              # noqa: E402


def main():
    ns = refimport.import_reference()
    cc, cm, ks = ns.cc, ns.cm, ns.ks
    S = 64
    cfg = ounet.UNetConfig(**ounet.TINY)
    sd = ounet.init_state_dict(cfg, seed=0)
    model, diffusion = build_ref_model(ns, ounet.TINY, sd)
    x0 = smooth_image(1, S, 1)
    kw = dict(in_shape=(1, 3, S, S), kernel_size=61, intensity=3.0, sigma_s=0.05)
    with refimport.reference_cwd():
        rop = cm.get_operator("gaussian_blur", device="cpu", **kw)
    oop = oops.get_operator("gaussian_blur", **kw)
    torch.manual_seed(2); meas_r = rop.forward(x0.clone(), flatten=True)
    torch.manual_seed(2); meas_o = oop.forward(x0.clone(), flatten=True)
    check("measurement", meas_r[0], meas_o[0], 1e-6)
    sig = ks.get_sigmas_karras(5, 0.01, 80, rho=7.0)
    xT = torch.randn(1, 3, S, S, generator=torch.Generator().manual_seed(3)) * 80
    rmodel = cc.ConditionOpenAIDenoiser(inner_model=model, diffusion=diffusion, x0_cov_type="convert", recon_mse=None, operator=rop,
                                        measurement=meas_r, guidance="I", mle_sigma_thres=0.2, device="cpu").eval()
    omodel = ocond.GuidedDenoiser(sd, cfg, oop, meas_o, "I", x0_cov_type="convert")
    trace_r = []
    x_r = ks.sample_dpmpp_2m(rmodel, xT.clone(), sig, disable=True, callback=lambda d: trace_r.append(d["denoised"].detach().clone()))
    x_o = osamp.sample_dpmpp_2m(omodel, xT.clone(), sig)
    check("sample_dpmpp_2m 5 steps", x_r, x_o, 5e-4)
    np.savez_compressed(os.path.join(GOLD, "sampler_dpmpp2m.npz"), sigmas=sig.numpy(), xT=xT.numpy(), x0=x_r.detach().numpy(),
                        denoised_first=trace_r[0].numpy(), denoised_last=trace_r[-1].numpy(), y=meas_r[0].numpy(), y_flat=meas_r[1].numpy())
    print("written", os.path.join(GOLD, "sampler_dpmpp2m.npz"))

    # ---- box / extreme masks: numpy seed -> mask bits (bit-exact contract, like the random mask)
    dump = {}
    for mt in ("box", "extreme"):
        for seed, S_, rng_ in ((0, 256, (128, 129)), (5, 256, (64, 160)), (7, 64, (16, 40))):
            opt = dict(mask_type=mt, mask_len_range=rng_, image_size=S_)
            with refimport.reference_cwd():
                np.random.seed(seed)
                r = cm.get_operator("inpainting", device="cpu", sigma_s=0.05, mask_opt=opt)
            np.random.seed(seed)
            o = oops.get_operator("inpainting", sigma_s=0.05, mask_opt=opt)
            assert torch.equal(r.mask, o.mask), (mt, seed)
            dump[f"{mt}|{seed}|{S_}|{rng_[0]}|{rng_[1]}"] = np.packbits(r.mask[0, 0].numpy().astype(np.uint8))
    np.savez_compressed(os.path.join(GOLD, "masks_box.npz"), **dump)
    print("written", os.path.join(GOLD, "masks_box.npz"), len(dump), "masks")


if __name__ == "__main__":
    main()
