"""Golden fixture for the caller-harness parity test: the ORACLE (torch-CPU restatement of the reference,
validated against the imported reference by make_golden.py) runs the harness' own protocol -- same seeded image, same operator
draw order, measurement noise and x_T from torch's CPU generator with the harness' seeds, 4-step Heun / Euler `--ode` runs --
and the PSNR / SSIM of the reference's compute_metrics are computed here with independent restatements (oracle.sampling.psnr,
a scipy.ndimage SSIM).  Output: tests/golden/harness_tiny.npz (inputs, expected sample, expected metrics).

Test infrastructure: only tests/ read the fixture.  usage: python oracle/make_golden_harness.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet as ounet, operators as oops, condition as ocond, sampling as osamp      # noqa: E402
from oracle.tables import get_sigmas_karras                                                       # noqa: E402
import kdip_amd.unet as ku                                                                        # noqa: E402  (synthetic weights only: CPU torch)
from sample_condition import synthetic_images                                                     # noqa: E402  (the harness' seeded image)


def ssim_scipy(a, b, R=1.0, win=7):
    """skimage.metrics.structural_similarity(channel_axis=0, data_range=1) restated on scipy.ndimage.uniform_filter."""
    from scipy.ndimage import uniform_filter
    out = []
    for x, y in zip(a.double().numpy(), b.double().numpy()):
        NP = win * win
        cn = NP / (NP - 1.0)
        ux, uy = uniform_filter(x, win), uniform_filter(y, win)
        vx = cn * (uniform_filter(x * x, win) - ux * ux)
        vy = cn * (uniform_filter(y * y, win) - uy * uy)
        vxy = cn * (uniform_filter(x * y, win) - ux * uy)
        C1, C2 = (0.01 * R) ** 2, (0.03 * R) ** 2
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2))
        p = (win - 1) // 2
        out.append(S[p:-p, p:-p].mean())
    return float(np.mean(out))


def main():
    seed, steps, S = 0, 4, 64
    cfg = ounet.UNetConfig(**ounet.TINY)
    sd = ku.synthetic_state_dict(seed=seed, image_size=S, model_channels=32, num_res_blocks=1, attention_resolutions="32", channel_mult=(1, 2))
    x0 = next(iter(synthetic_images(1, S)))[None]
    dump = {"x0": x0.numpy()}
    tasks = {"gaussian_deblur_64": ("gaussian_blur", dict(in_shape=(1, 3, S, S), kernel_size=61, intensity=3.0, sigma_s=0.05), "I", "convert", {}),
             "inpainting_64": ("inpainting", dict(sigma_s=0.05, mask_opt=dict(mask_type="random", mask_prob_range=(0.5, 0.5), image_size=S)), "dps", "dps", dict(zeta=1.0))}
    for task, (name, kw, guidance, cov, extra) in tasks.items():
        for sampler in ("heun", "euler"):
            np.random.seed(seed)                                   # sample_condition.py: masks come from numpy's global stream
            op = oops.get_operator(name, **kw)
            torch.manual_seed(seed * 1000003 + 0)                  # measurement of image 0
            meas = op.forward(x0.clone(), flatten=True)
            torch.manual_seed(seed + 7919 * 1 + 0)                 # x_T of image 0 on rank 0
            xT = torch.randn(1, 3, S, S) * 80
            sig = get_sigmas_karras(steps, 0.01, 80)
            model = ocond.GuidedDenoiser(sd, cfg, op, meas, guidance, x0_cov_type=cov, zeta=extra.get("zeta"), mle_sigma_thres=0.2)
            fn = osamp.sample_heun if sampler == "heun" else osamp.sample_euler
            hat = fn(model, xT.clone(), sig)
            a, b = (x0[0] / 2 + 0.5).clip(0, 1), (hat[0] / 2 + 0.5).clip(0, 1)
            psnr = float(osamp.psnr(hat, x0)[0])
            ssim = ssim_scipy(a, b)
            key = f"{task}.{sampler}"
            dump[key + ".hat"] = hat.detach().numpy()
            dump[key + ".y"] = meas[0].detach().numpy()
            dump[key + ".psnr"] = np.float64(psnr)
            dump[key + ".ssim"] = np.float64(ssim)
            print(f"{key}: psnr {psnr:.6f} ssim {ssim:.6f}")
    out = os.path.join(ROOT, "tests", "golden", "harness_tiny.npz")
    np.savez_compressed(out, **dump)
    print("written", out)


if __name__ == "__main__":
    main()
