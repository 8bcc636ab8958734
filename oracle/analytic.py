"""Oracle: Monte-Carlo estimate of the denoiser MSE per sigma (analytic_variance.py:113-139).

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
import torch

from .tables import DiffusionTables
from .unet import unet_forward


def denoise(sd, cfg, x, sigma, D=None):
    """OpenAIDenoiser.forward (k_diffusion/external.py:108-112,115-130): x - sigma * eps(x*c_in, t_frac)."""
    D = D or DiffusionTables()
    s0 = sigma[:1]
    c_in = 1 / (s0 ** 2 + 1) ** 0.5
    t = D.sigma_to_t(sigma)
    eps = unet_forward(sd, cfg, x * c_in, t).chunk(2, dim=1)[0]
    return x + eps * (-s0)


@torch.no_grad()
def estimate_recon_mse(sd, cfg, batches, sigmas):
    """For every sigma: mean over batches of mean((x0 - D(x0 + sigma n; sigma))^2); noise from the global
    torch RNG in (sigma, batch) order."""
    mse_list, errors = [], torch.zeros(len(sigmas), len(batches))
    for i, sigma in enumerate(sigmas):
        mse = 0
        for j, x0 in enumerate(batches):
            hat = denoise(sd, cfg, x0 + torch.randn_like(x0) * sigma, sigma.repeat(x0.shape[0]))
            cur = (x0 - hat).pow(2).mean()
            errors[i, j] = cur
            mse = mse + cur
        mse_list.append(mse / len(batches))
    return {"sigmas": sigmas, "mse_list": torch.stack(mse_list), "errors": errors}
