"""DEV-ONLY: a MID-SIZE fixture captured from the real reference, sized so that the production bf16 kernels of the large-map path
(csrc/conv3.hip: GroupNorm + SiLU staging, fused forward / backward statistics, GroupNorm-backward staging, coefficient fold) run
against reference output and not only against the oracle: model_channels = 128, 64 x 64, channel_mult (1, 2), attention at 32 x 32,
batch 12 (12 x 8 x 2 = 192 conv3 tiles per 128-channel launch at 64 x 64 = the routing threshold of csrc/unet.hip).

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference).  Usage: python -m oracle.make_golden_mid
Writes tests/golden/unet_mid.npz: expected outputs only -- the inputs are re-drawn from the same CPU seeds by the test
(x: seed 41, t: fixed list, cotangent: seed 42; guided-call inputs: x0 + sigma * randn(seed 100 + i)) -- after asserting
reference == oracle:
  out, vjp          UNetModel.forward (guided_diffusion/unet.py:636-668) and its input-VJP (autograd) at batch 12
  hat|<sigma>       12 batch-1 ConditionOpenAIDenoiser calls (condition/condition.py:83-131; Gaussian deblur, Type-I, Convert) per sigma
"""
import os
import sys
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refimport, unet as ounet, operators as oops, condition as ocond      # noqa: E402
from oracle.make_golden import build_ref_model, smooth_image, check, GOLD               # noqa: E402

MID = dict(image_size=64, model_channels=128, num_res_blocks=1, attention_resolutions="32", channel_mult=(1, 2))
B = 12
T_LIST = [3.0, 57.0, 120.0, 258.0, 333.0, 401.0, 512.0, 673.0, 740.0, 801.0, 929.0, 999.0]


def inputs():
    x = torch.randn(B, 3, 64, 64, generator=torch.Generator().manual_seed(41))
    cot = torch.randn(B, 6, 64, 64, generator=torch.Generator().manual_seed(42))
    return x, torch.tensor(T_LIST), cot


def guided_inputs(sigma_v):
    x0 = smooth_image(B, 64, seed=43)
    return x0, torch.cat([x0[i:i + 1] + sigma_v * torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(100 + i)) for i in range(B)])


def main():
    torch.set_num_threads(8)
    ns = refimport.import_reference()
    cc, cm = ns.cc, ns.cm
    cfg = ounet.UNetConfig(**MID)
    sd = ounet.init_state_dict(cfg, seed=0)
    model, diffusion = build_ref_model(ns, MID, sd)
    x, t, cot = inputs()
    with torch.no_grad():
        o_ref = model(x, t)
        o_ora = ounet.unet_forward(sd, cfg, x, t)
    check("mid UNetModel.forward", o_ref, o_ora, 5e-5)
    xr = x.clone().requires_grad_()
    g_ref = torch.autograd.grad((model(xr, t) * cot).sum(), xr)[0]
    xo = x.clone().requires_grad_()
    g_ora = torch.autograd.grad((ounet.unet_forward(sd, cfg, xo, t) * cot).sum(), xo)[0]
    check("mid UNet input-VJP", g_ref, g_ora, 5e-5 * max(1.0, float(g_ref.abs().max())))
    dump = {"out": o_ref.numpy(), "vjp": g_ref.numpy()}
    kw = dict(in_shape=(1, 3, 64, 64), kernel_size=61, intensity=3.0, sigma_s=0.05)
    with refimport.reference_cwd():
        rop = cm.get_operator("gaussian_blur", device="cpu", **kw)
    oop = oops.get_operator("gaussian_blur", **kw)
    for sigma_v in (1.5, 0.12):
        x0, xs = guided_inputs(sigma_v)
        hats = []
        for i in range(B):
            torch.manual_seed(2 + i); meas_r = rop.forward(x0[i:i + 1].clone(), flatten=True)
            torch.manual_seed(2 + i); meas_o = oop.forward(x0[i:i + 1].clone(), flatten=True)
            rmodel = cc.ConditionOpenAIDenoiser(inner_model=model, diffusion=diffusion, x0_cov_type="convert", recon_mse=None, operator=rop,
                                                measurement=meas_r, guidance="I", mle_sigma_thres=0.2, device="cpu").eval()
            omodel = ocond.GuidedDenoiser(sd, cfg, oop, meas_o, "I", x0_cov_type="convert")
            h_r = rmodel(xs[i:i + 1].clone(), torch.tensor([sigma_v]))
            h_o = omodel(xs[i:i + 1].clone(), torch.tensor([sigma_v]))
            check(f"mid guided call sigma={sigma_v} image {i}", h_r, h_o, 5e-4)
            hats.append(h_r.detach())
        dump[f"hat|{sigma_v}"] = torch.cat(hats).numpy()
    np.savez_compressed(os.path.join(GOLD, "unet_mid.npz"), **dump)
    print("written", os.path.join(GOLD, "unet_mid.npz"))


if __name__ == "__main__":
    main()
