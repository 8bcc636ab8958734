"""Karras schedule + Euler / Heun samplers -- same call surface as k_diffusion/sampling.py
(get_sigmas_karras :17-23, to_d :46-48, sample_euler :118-135, sample_heun :159-184).

`model(x, sigma * s_in, **extra_args)` must return the denoised image (same shape / dtype /
device).  The per-step tensor updates run as fused HIP kernels (kdip_sampler_*); the scalar
schedule arithmetic is done in fp32 on the host exactly as the reference does it on 0-d
tensors, so no device sync is needed inside the loop.  Random draws stay on torch's RNG
(`torch.randn_like`), one per step even when gamma == 0, to keep the stream identical.
"""
import torch

from . import _lib as L


def append_zero(x):
    return torch.cat([x, x.new_zeros([1])])


def get_sigmas_karras(n, sigma_min, sigma_max, rho=7., device='cpu'):
    """Noise schedule of Karras et al. (2022)."""
    ramp = torch.linspace(0, 1, n)
    min_inv_rho = sigma_min ** (1 / rho)
    max_inv_rho = sigma_max ** (1 / rho)
    sigmas = (max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** rho
    return append_zero(sigmas).to(device)


def _sigma_vec(x, value):
    """sigma * s_in with the host value attached so the denoiser needs no .item() sync."""
    s = x.new_full([x.shape[0]], float(value))
    s._kdip_host_value = float(value)
    return s


def to_d(x, sigma, denoised):
    """Karras ODE derivative (x - denoised) / sigma."""
    out = torch.empty_like(x)
    L.check(L.load().kdip_axpby(L.stream(), L.ptr(x.contiguous()), 1.0 / float(sigma), L.ptr(denoised.contiguous()),
                                -1.0 / float(sigma), x.numel(), L.ptr(out)))
    return out


def _prep(x, sigmas):
    L.require_gpu()
    if not (x.is_cuda and x.dtype == torch.float32):
        raise L.KdipError("kdip_amd samplers need a float32 CUDA(HIP) tensor")
    return x.contiguous(), sigmas.detach().to("cpu", torch.float32)


def _trange(n, disable):
    try:
        from tqdm.auto import trange
        return trange(n, disable=disable)
    except Exception:
        return range(n)


def _churn(x, sig, i, s_churn, s_tmin, s_tmax, s_noise, lib, noise_fn=None):
    gamma = min(s_churn / (len(sig) - 1), 2 ** 0.5 - 1) if s_tmin <= sig[i] <= s_tmax else 0.
    # noise_fn(x) -> eps: lets a caller inject pre-drawn noise (e.g. the reference's CPU stream) instead of the device RNG
    eps = torch.randn_like(x) if noise_fn is None else noise_fn(x).to(x.device, x.dtype).contiguous()
    sigma_hat = sig[i] * (gamma + 1)
    if gamma > 0:
        scale = float((sigma_hat ** 2 - sig[i] ** 2) ** 0.5) * float(s_noise)
        xn = torch.empty_like(x)
        L.check(lib.kdip_sampler_add_noise(L.stream(), L.ptr(x), L.ptr(eps), scale, x.numel(), L.ptr(xn)))
        x = xn
    return x, sigma_hat


def sample_euler(model, x, sigmas, extra_args=None, callback=None, disable=None, s_churn=0., s_tmin=0.,
                 s_tmax=float('inf'), s_noise=1., noise_fn=None):
    """Algorithm 2 (Euler steps) from Karras et al. (2022)."""
    extra_args = {} if extra_args is None else extra_args
    lib = L.load()
    x, sig = _prep(x, sigmas)
    for i in _trange(len(sig) - 1, disable):
        x, sigma_hat = _churn(x, sig, i, s_churn, s_tmin, s_tmax, s_noise, lib, noise_fn)
        denoised = model(x, _sigma_vec(x, sigma_hat), **extra_args).contiguous()
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': sigmas[i], 'sigma_hat': sigma_hat, 'denoised': denoised})
        dt = sig[i + 1] - sigma_hat
        xn = torch.empty_like(x)
        L.check(lib.kdip_sampler_euler(L.stream(), L.ptr(x), L.ptr(denoised), float(sigma_hat), float(dt), x.numel(), L.ptr(xn)))
        x = xn
    return x


def heun_step(model, x, sig, i, extra_args=None, callback=None, sigmas=None, s_churn=0., s_tmin=0.,
              s_tmax=float('inf'), s_noise=1., noise_fn=None):
    """One iteration i of sample_heun's loop on the host-side fp32 schedule `sig` (2 model calls,
    1 for the last step).  Exposed so a harness can time individual sampler steps."""
    extra_args = {} if extra_args is None else extra_args
    lib = L.load()
    n = x.numel()
    x, sigma_hat = _churn(x, sig, i, s_churn, s_tmin, s_tmax, s_noise, lib, noise_fn)
    denoised = model(x, _sigma_vec(x, sigma_hat), **extra_args).contiguous()
    if callback is not None:
        callback({'x': x, 'i': i, 'sigma': (sigmas if sigmas is not None else sig)[i], 'sigma_hat': sigma_hat,
                  'denoised': denoised})
    dt = sig[i + 1] - sigma_hat
    xn = torch.empty_like(x)
    if sig[i + 1] == 0:
        L.check(lib.kdip_sampler_euler(L.stream(), L.ptr(x), L.ptr(denoised), float(sigma_hat), float(dt), n, L.ptr(xn)))
        return xn
    x_2 = torch.empty_like(x)
    L.check(lib.kdip_sampler_euler(L.stream(), L.ptr(x), L.ptr(denoised), float(sigma_hat), float(dt), n, L.ptr(x_2)))
    denoised_2 = model(x_2, _sigma_vec(x, sig[i + 1]), **extra_args).contiguous()
    L.check(lib.kdip_sampler_heun(L.stream(), L.ptr(x), L.ptr(denoised), L.ptr(x_2), L.ptr(denoised_2),
                                  float(sigma_hat), float(sig[i + 1]), float(dt), n, L.ptr(xn)))
    return xn


def sample_heun(model, x, sigmas, extra_args=None, callback=None, disable=None, s_churn=0., s_tmin=0.,
                s_tmax=float('inf'), s_noise=1., noise_fn=None):
    """Algorithm 2 (Heun steps) from Karras et al. (2022); the last step (sigma -> 0) is Euler."""
    x, sig = _prep(x, sigmas)
    for i in _trange(len(sig) - 1, disable):
        x = heun_step(model, x, sig, i, extra_args, callback, sigmas, s_churn, s_tmin, s_tmax, s_noise, noise_fn)
    return x


def sample_dpmpp_2m(model, x, sigmas, extra_args=None, callback=None, disable=None):
    """DPM-Solver++(2M) (k_diffusion/sampling.py:583-605; the training-preview sampler, train_openai.py:114).
    The log-sigma step arithmetic is fp32 on the host, as the reference does it on 0-d tensors; the update
    x' = (sigma_next / sigma) x - expm1(-h) * ((1 + 1/2r) denoised - (1/2r) old_denoised) is two fused axpby kernels."""
    extra_args = {} if extra_args is None else extra_args
    lib = L.load()
    x, sig = _prep(x, sigmas)
    t_fn = lambda s: s.log().neg()
    sigma_fn = lambda t: t.neg().exp()
    old_denoised = None
    for i in _trange(len(sig) - 1, disable):
        denoised = model(x, _sigma_vec(x, sig[i]), **extra_args).contiguous()
        if callback is not None:
            callback({'x': x, 'i': i, 'sigma': sigmas[i], 'sigma_hat': sigmas[i], 'denoised': denoised})
        t, t_next = t_fn(sig[i]), t_fn(sig[i + 1])
        h = t_next - t
        a = float(sigma_fn(t_next) / sigma_fn(t))
        e = float((-h).expm1())
        xn = torch.empty_like(x)
        if old_denoised is None or sig[i + 1] == 0:
            L.check(lib.kdip_axpby(L.stream(), L.ptr(x), a, L.ptr(denoised), -e, x.numel(), L.ptr(xn)))
        else:
            r = (t - t_fn(sig[i - 1])) / h
            c1, c2 = float(1 + 1 / (2 * r)), float(1 / (2 * r))
            tmp = torch.empty_like(x)
            L.check(lib.kdip_axpby(L.stream(), L.ptr(x), a, L.ptr(denoised), -e * c1, x.numel(), L.ptr(tmp)))
            L.check(lib.kdip_axpby(L.stream(), L.ptr(tmp), 1.0, L.ptr(old_denoised), e * c2, x.numel(), L.ptr(xn)))
        x, old_denoised = xn, denoised
    return x
