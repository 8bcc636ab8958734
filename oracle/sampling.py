"""Oracle: Karras Euler/Heun samplers (k_diffusion/sampling.py:46-48,118-135,159-184).

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
import torch

from .tables import get_sigmas_karras  # noqa: F401  (re-export, sampling.py:17-23)


def to_d(x, sigma, denoised):
    """sampling.py:46-48 (sigma is a 0-d tensor here)."""
    return (x - denoised) / sigma


def _gamma(sigmas, i, s_churn, s_tmin, s_tmax):
    return min(s_churn / (len(sigmas) - 1), 2 ** 0.5 - 1) if s_tmin <= sigmas[i] <= s_tmax else 0.


def sample_euler(model, x, sigmas, callback=None, s_churn=0., s_tmin=0., s_tmax=float("inf"), s_noise=1.):
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        gamma = _gamma(sigmas, i, s_churn, s_tmin, s_tmax)
        eps = torch.randn_like(x) * s_noise
        sigma_hat = sigmas[i] * (gamma + 1)
        if gamma > 0:
            x = x + eps * (sigma_hat ** 2 - sigmas[i] ** 2) ** 0.5
        denoised = model(x, sigma_hat * s_in)
        d = to_d(x, sigma_hat, denoised)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigma_hat, "denoised": denoised})
        dt = sigmas[i + 1] - sigma_hat
        x = x + d * dt
    return x


def sample_heun(model, x, sigmas, callback=None, s_churn=0., s_tmin=0., s_tmax=float("inf"), s_noise=1.):
    s_in = x.new_ones([x.shape[0]])
    for i in range(len(sigmas) - 1):
        gamma = _gamma(sigmas, i, s_churn, s_tmin, s_tmax)
        eps = torch.randn_like(x) * s_noise
        sigma_hat = sigmas[i] * (gamma + 1)
        if gamma > 0:
            x = x + eps * (sigma_hat ** 2 - sigmas[i] ** 2) ** 0.5
        denoised = model(x, sigma_hat * s_in)
        d = to_d(x, sigma_hat, denoised)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigma_hat, "denoised": denoised})
        dt = sigmas[i + 1] - sigma_hat
        if sigmas[i + 1] == 0:
            x = x + d * dt
        else:
            x_2 = x + d * dt
            denoised_2 = model(x_2, sigmas[i + 1] * s_in)
            d_2 = to_d(x_2, sigmas[i + 1], denoised_2)
            x = x + (d + d_2) / 2 * dt
    return x


def psnr(hat_x0, x0):
    """compute_metrics PSNR (sample_condition_openai.py:41-44): 10 log10(1/MSE) on
    (x/2+0.5).clip(0,1), data_range 1."""
    a = (hat_x0 / 2 + 0.5).clip(0, 1)
    b = (x0 / 2 + 0.5).clip(0, 1)
    mse = ((a - b) ** 2).flatten(1).mean(dim=1)
    return 10 * torch.log10(1.0 / mse)


def sample_dpmpp_2m(model, x, sigmas, callback=None):
    """DPM-Solver++(2M), k_diffusion/sampling.py:583-605 (used by the training preview, train_openai.py:114)."""
    s_in = x.new_ones([x.shape[0]])
    t_fn = lambda sigma: sigma.log().neg()
    sigma_fn = lambda t: t.neg().exp()
    old_denoised = None
    for i in range(len(sigmas) - 1):
        denoised = model(x, sigmas[i] * s_in)
        if callback is not None:
            callback({"x": x, "i": i, "sigma": sigmas[i], "sigma_hat": sigmas[i], "denoised": denoised})
        t, t_next = t_fn(sigmas[i]), t_fn(sigmas[i + 1])
        h = t_next - t
        if old_denoised is None or sigmas[i + 1] == 0:
            x = (sigma_fn(t_next) / sigma_fn(t)) * x - (-h).expm1() * denoised
        else:
            h_last = t - t_fn(sigmas[i - 1])
            r = h_last / h
            denoised_d = (1 + 1 / (2 * r)) * denoised - (1 / (2 * r)) * old_denoised
            x = (sigma_fn(t_next) / sigma_fn(t)) * x - (-h).expm1() * denoised_d
        old_denoised = denoised
    return x
