"""Oracle: DDPM tables, sigma<->t mapping, timestep embedding, Karras schedule.

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
import math
import numpy as np
import torch


class DiffusionTables:
    """float64 tables of GaussianDiffusion.__init__
    (guided_diffusion/gaussian_diffusion.py:118-169) for the linear schedule
    (`get_named_beta_schedule`, :27-35), no respacing (respace.py:63-91 with
    use_timesteps = all => timestep_map = identity)."""

    def __init__(self, steps=1000):
        scale = 1000 / steps
        base_betas = np.linspace(scale * 0.0001, scale * 0.02, steps, dtype=np.float64)
        # SpacedDiffusion re-derives betas from the base alphas_cumprod even with no
        # respacing (respace.py:71-80): 1 - ac[i]/ac[i-1]; differs from base in the last ulp.
        base_ac = np.cumprod(1.0 - base_betas, axis=0)
        last = 1.0
        new_betas = []
        for ac in base_ac:
            new_betas.append(1 - ac / last)
            last = ac
        betas = np.array(new_betas, dtype=np.float64)
        self.num_timesteps = steps
        self.betas = betas
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(
            np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.log_betas = np.log(betas)

    def f32(self, arr, t):
        """_extract_into_tensor: index float64 then .float()
        (gaussian_diffusion.py:895-908)."""
        return torch.from_numpy(arr)[t].float()

    # ---- sigma grid of DiscreteEpsDDPMDenoiser (k_diffusion/external.py:93,121) ----
    def sigma_grid(self):
        ac = torch.tensor(self.alphas_cumprod, dtype=torch.float32)
        return ((1 - ac) / ac) ** 0.5

    def sigma_to_t(self, sigma):
        """DiscreteSchedule.sigma_to_t, quantize=False (k_diffusion/external.py:67-79).
        sigma: fp32 tensor [B]; returns fractional t fp32 [B]."""
        log_sigmas = self.sigma_grid().log()
        log_sigma = sigma.log()
        dists = log_sigma - log_sigmas[:, None]
        low_idx = dists.ge(0).cumsum(dim=0).argmax(dim=0).clamp(max=log_sigmas.shape[0] - 2)
        high_idx = low_idx + 1
        low, high = log_sigmas[low_idx], log_sigmas[high_idx]
        w = ((low - log_sigma) / (low - high)).clamp(0, 1)
        t = (1 - w) * low_idx + w * high_idx
        return t.view(sigma.shape)


def timestep_embedding(timesteps, dim, max_period=10000):
    """guided_diffusion/nn.py:103-121."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def get_sigmas_karras(n, sigma_min, sigma_max, rho=7.0):
    """k_diffusion/sampling.py:17-23."""
    ramp = torch.linspace(0, 1, n)
    min_inv_rho = sigma_min ** (1 / rho)
    max_inv_rho = sigma_max ** (1 / rho)
    sigmas = (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho
    return torch.cat([sigmas, sigmas.new_zeros([1])])
