"""sigma-space wrappers -- call surface of k_diffusion/external.py for the OpenAI (ADM)
models: DiscreteSchedule (:42-85), DiscreteEpsDDPMDenoiser (:88-112), OpenAIDenoiser
(:115-130), OpenAIDenoiserV2 (:133-169).  Schedule maths is host-side fp32 torch (tiny,
and must reproduce the reference's rounding: sigma grid computed in fp32, external.py:93,121);
tensor work goes through libkdip_hip.
"""
import torch

from . import _lib as L
from .transforms import OrthoTransform


def sigma_host(sigma):
    """float value of a per-call sigma tensor (all entries equal) without a device sync when
    the sampler attached it."""
    v = getattr(sigma, "_kdip_host_value", None)
    if v is not None:
        return float(torch.tensor(v, dtype=torch.float32))
    return float(sigma.reshape(-1)[0])


class DiscreteSchedule:
    """Mapping between continuous noise levels and the model's discrete levels."""

    def __init__(self, sigmas, quantize):
        self.sigmas = sigmas.detach().to("cpu", torch.float32)
        self.log_sigmas = self.sigmas.log()
        self.quantize = quantize

    @property
    def sigma_min(self):
        return self.sigmas[0]

    @property
    def sigma_max(self):
        return self.sigmas[-1]

    def sigma_to_t(self, sigma, quantize=None):
        quantize = self.quantize if quantize is None else quantize
        dev = sigma.device
        sigma_c = sigma.detach().to("cpu", torch.float32) if getattr(sigma, "_kdip_host_value", None) is None \
            else torch.full(sigma.shape, sigma._kdip_host_value, dtype=torch.float32)
        log_sigma = sigma_c.log()
        dists = log_sigma - self.log_sigmas[:, None]
        if quantize:
            return dists.abs().argmin(dim=0).view(sigma.shape).to(dev)
        low_idx = dists.ge(0).cumsum(dim=0).argmax(dim=0).clamp(max=self.log_sigmas.shape[0] - 2)
        high_idx = low_idx + 1
        low, high = self.log_sigmas[low_idx], self.log_sigmas[high_idx]
        w = ((low - log_sigma) / (low - high)).clamp(0, 1)
        t = (1 - w) * low_idx + w * high_idx
        return t.view(sigma.shape)          # host tensor: consumers pass it to the UNet handle

    def t_to_sigma(self, t):
        t = t.float().cpu()
        low_idx, high_idx, w = t.floor().long(), t.ceil().long(), t.frac()
        log_sigma = (1 - w) * self.log_sigmas[low_idx] + w * self.log_sigmas[high_idx]
        return log_sigma.exp()


class DiscreteEpsDDPMDenoiser(DiscreteSchedule):
    """Wrapper for discrete-schedule DDPM models that output eps."""

    def __init__(self, model, alphas_cumprod, quantize):
        super().__init__(((1 - alphas_cumprod) / alphas_cumprod) ** 0.5, quantize)
        self.inner_model = model
        self.sigma_data = 1.

    def get_scalings(self, sigma):
        c_out = -sigma
        c_in = 1 / (sigma ** 2 + self.sigma_data ** 2) ** 0.5
        return c_out, c_in

    def eval(self):
        return self


class OpenAIDenoiser(DiscreteEpsDDPMDenoiser):
    """OpenAI diffusion model -> Karras denoiser D(x; sigma) = x - sigma * eps(x * c_in, t)."""

    def __init__(self, model, diffusion, quantize=False, has_learned_sigmas=True, device='cpu'):
        alphas_cumprod = torch.tensor(diffusion.alphas_cumprod, dtype=torch.float32)
        super().__init__(model, alphas_cumprod, quantize=quantize)
        self.has_learned_sigmas = has_learned_sigmas

    def forward(self, input, sigma):
        s = sigma_host(sigma)
        s32 = torch.tensor(s, dtype=torch.float32)
        c_in = float(1 / (s32 ** 2 + 1) ** 0.5)
        t = self.sigma_to_t(sigma)
        out, _, _ = self.inner_model.forward_raw(input, t, in_scale=c_in)
        eps = out[:, :3].contiguous()
        res = torch.empty_like(input)
        L.check(L.load().kdip_axpby(L.stream(), L.ptr(input.contiguous()), 1.0, L.ptr(eps), -s, input.numel(), L.ptr(res)))
        return res

    def loss(self, input, noise, sigma):
        """eps-prediction objective inherited from DiscreteEpsDDPMDenoiser (k_diffusion/external.py:105-109), forward value:
        per sample mean[(eps_hat(x + sigma*noise) - noise)^2].  `sigma` [B]: samples that share a sigma share one UNet call."""
        lib = L.load()
        B = input.shape[0]
        sig = sigma.detach().to("cpu", torch.float32).reshape(-1)
        out = torch.empty(B, device=input.device)
        per = input[0].numel()
        for s in sorted(set(sig.tolist())):
            idx = [i for i in range(B) if float(sig[i]) == s]
            x = input[idx].contiguous()
            n = noise[idx].contiguous()
            xn = torch.empty_like(x)
            L.check(lib.kdip_axpby(L.stream(), L.ptr(x), 1.0, L.ptr(n), float(s), x.numel(), L.ptr(xn)))
            sv = x.new_full([len(idx)], float(s))
            sv._kdip_host_value = float(s)
            s32 = torch.tensor(float(s), dtype=torch.float32)
            c_in = float(1 / (s32 ** 2 + 1) ** 0.5)
            o, _, _ = self.inner_model.forward_raw(xn, self.sigma_to_t(sv), in_scale=c_in)
            eps = (o[:, :3] if self.has_learned_sigmas else o).contiguous()
            zero = torch.zeros_like(eps)              # unit variance: the NLL reduction kernel then returns mean (eps - noise)^2
            part = torch.empty(len(idx), device=input.device)
            L.check(lib.kdip_gauss_nll_mean(L.stream(), L.ptr(eps), L.ptr(n), L.ptr(zero), len(idx), per, 0, L.ptr(part)))
            out[idx] = part
        return out

    __call__ = forward


class OpenAIDenoiserV2(DiscreteEpsDDPMDenoiser):
    """OpenAI model + `out_cov` 1x1 head predicting log-variances in pixel and transform space
    (k_diffusion/external.py:133-169).  The head's weights live in the UNet handle
    (state_dict keys out_cov.weight / out_cov.bias)."""

    def __init__(self, model, diffusion, quantize=False, device='cpu', ortho_tf_type=None):
        alphas_cumprod = torch.tensor(diffusion.alphas_cumprod, dtype=torch.float32)
        super().__init__(model, alphas_cumprod, quantize=quantize)
        self.ortho_tf_type = ortho_tf_type
        self.ortho_tf = OrthoTransform(ortho_tf_type)

    def forward(self, input, sigma, return_variance=False):
        s = sigma_host(sigma)
        s32 = torch.tensor(s, dtype=torch.float32)
        c_in = float(1 / (s32 ** 2 + 1) ** 0.5)
        t = self.sigma_to_t(sigma)                                  # fractional, NOT floored (external.py:163)
        out, cov, _ = self.inner_model.forward_raw(input, t, in_scale=c_in, want_cov=True)
        model_output = out[:, :3]
        if return_variance:
            logvar, logvar_ot = cov.chunk(2, dim=1)
            return model_output, logvar, logvar_ot
        res = torch.empty_like(input)
        L.check(L.load().kdip_axpby(L.stream(), L.ptr(input.contiguous()), 1.0, L.ptr(model_output.contiguous()), -s,
                                    input.numel(), L.ptr(res)))
        return res

    def loss(self, input, noise, sigma):
        """The DWT-Var / DCT-Var objective (k_diffusion/external.py:145-159), evaluated forward-only (training itself is out of
        scope): per sample  mean[(eps_hat - eps*)^2 / exp(logvar) + logvar] + the same in the transform basis with logvar_ot,
        where eps* = (x - x_noised) / c_out = noise.  `sigma` [B]: samples that share a sigma share one UNet call."""
        lib = L.load()
        B = input.shape[0]
        sig = sigma.detach().to("cpu", torch.float32).reshape(-1)
        out = torch.empty(B, device=input.device)
        per = input[0].numel()
        for s in sorted(set(sig.tolist())):
            idx = [i for i in range(B) if float(sig[i]) == s]
            x = input[idx].contiguous()
            n = noise[idx].contiguous()
            xn = torch.empty_like(x)
            L.check(lib.kdip_axpby(L.stream(), L.ptr(x), 1.0, L.ptr(n), float(s), x.numel(), L.ptr(xn)))
            sv = x.new_full([len(idx)], float(s))
            sv._kdip_host_value = float(s)
            model_output, logvar, logvar_ot = self.forward(xn, sv, return_variance=True)
            model_output, logvar, logvar_ot = model_output.contiguous(), logvar.contiguous(), logvar_ot.contiguous()
            # target = (input - noised_input) / c_out with c_out = -sigma  ==  noise (kept in the reference's form for the rounding)
            target = torch.empty_like(x)
            L.check(lib.kdip_axpby(L.stream(), L.ptr(x), -1.0 / float(s), L.ptr(xn), 1.0 / float(s), x.numel(), L.ptr(target)))
            part = torch.empty(len(idx), device=input.device)
            L.check(lib.kdip_gauss_nll_mean(L.stream(), L.ptr(model_output), L.ptr(target), L.ptr(logvar), len(idx), per, 0, L.ptr(part)))
            ot = self.ortho_tf
            mo_t, tg_t = ot(model_output).contiguous(), ot(target).contiguous()      # (named: raw pointers must outlive the call)
            L.check(lib.kdip_gauss_nll_mean(L.stream(), L.ptr(mo_t), L.ptr(tg_t), L.ptr(logvar_ot), len(idx), per, 1, L.ptr(part)))
            out[idx] = part
        return out

    __call__ = forward
