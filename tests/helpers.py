"""Shared test helpers (seeded synthetic inputs)."""
import numpy as np
import torch


def smooth_image(B, size, seed=1):
    g = torch.Generator().manual_seed(seed)
    r = torch.rand(B, 3, size, size, generator=g) * 2 - 1
    rp = torch.nn.functional.pad(r, (4, 4, 4, 4), mode="circular")
    return (3 * torch.nn.functional.avg_pool2d(rp, 9, 1)).clamp(-1, 1)


def op_cfgs(S):
    return {
        "gaussian_blur": dict(in_shape=(1, 3, S, S), kernel_size=61, intensity=3.0, sigma_s=0.05),
        "motion_blur": dict(in_shape=(1, 3, S, S), kernel_size=61, intensity=0.5, sigma_s=0.05),
        "super_resolution": dict(in_shape=(1, 3, S, S), scale_factor=4, sigma_s=0.05),
        "inpainting": dict(sigma_s=0.05, mask_opt=dict(mask_type="random", mask_prob_range=(0.5, 0.5), image_size=S)),
    }


def synthetic_recon_mse():
    from oracle.tables import get_sigmas_karras
    s = get_sigmas_karras(1000, 0.01, 80)[:-1]
    return {"sigmas": s, "mse_list": s ** 2 / (1 + s ** 2) * 0.5}


def psnr_db(a, b, data_range=2.0):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 10 * np.log10(data_range ** 2 / max(mse, 1e-30))
