"""Analytic posterior-variance table -- the estimator of analytic_variance.py:113-139
(`--xstart-cov-type analytic` consumes its output, condition.py:250-254).

For every sigma of a Karras schedule: Monte-Carlo mean over a data subset of
mean((x0 - D(x0 + sigma n; sigma))^2), with D the unguided OpenAIDenoiser (fractional t, no clamp).
Forward-only UNet passes through libkdip_hip; the final scalar mean uses torch (offline tool, not
on the sampler path).  Returns the reference's dict layout {'sigmas', 'mse_list', 'errors'}.
"""
import torch

from .external import OpenAIDenoiser
from .sampling import get_sigmas_karras, _sigma_vec


@torch.no_grad()
def estimate_recon_mse(inner_model, diffusion, batches, sigmas=None, steps=1000, sigma_min=0.01, sigma_max=80.0):
    model = OpenAIDenoiser(inner_model, diffusion)
    batches = list(batches)
    dev = batches[0].device
    if sigmas is None:
        sigmas = get_sigmas_karras(steps, sigma_min, sigma_max, rho=7., device=dev)
    sig_host = sigmas.detach().cpu()
    mse_list, errors = [], torch.zeros(len(sig_host), len(batches))
    for i, sigma in enumerate(sig_host):
        mse = 0.0
        for j, x0 in enumerate(batches):
            xt = x0 + torch.randn_like(x0) * float(sigma)
            hat = model(xt.contiguous(), _sigma_vec(x0, sigma)) if float(sigma) > 0 else xt
            cur = float((x0 - hat).pow(2).mean())
            errors[i, j] = cur
            mse += cur
        mse_list.append(mse / len(batches))
    return {"sigmas": sig_host, "mse_list": torch.tensor(mse_list), "errors": errors}
