"""Guided denoiser E[x0 | x_t, y] -- plug-in surface of condition/condition.py:
ConditionDenoiser (:41-208), ConditionOpenAIDenoiser (:211-274), ConditionOpenAIDenoiserV2
(:277-300), the mat-solver registry (:307-314) and the four built-in solvers (:317-439).

Same constructor kwargs, guidance strings and error behaviour; `forward(x, sigma)` returns the
detached, clipped x0 estimate.  The batch-1 restriction of the reference is lifted: B
independent problems that share one sigma per call (per-sample norms and CG).

Where the reference differentiates through the UNet with autograd, this module calls the
hand-written input-VJP of libkdip_hip:
    V1:  x0 = clamp(a_t * x*c_in - b_t * eps(x*c_in, t))     a_t = sqrt(1/ac_t), b_t = sqrt(1/ac_t - 1)
         (dx0/dx)^T g = c_in * (a_t * g_raw + UNet^T(-b_t * g_raw (+) 0)),  g_raw = g * 1[|x0_raw| <= 1]
    V2:  x0 = x - sigma * eps(x*c_in, t)
         (dx0/dx)^T g = g - sigma * c_in * UNet^T(g (+) 0)
"""
from abc import abstractmethod
from warnings import warn

import torch
from torch import nn

from . import _lib as L
from .external import OpenAIDenoiser, OpenAIDenoiserV2, sigma_host
from .transforms import OrthoTransform


class ConditionDenoiser(nn.Module):
    '''Approximate E[x0|xt, y] given variational Gaussian posterior'''

    def __init__(self, operator, measurement, guidance, device='cpu', zeta=None, lambda_=None, eta=None,
                 num_hutchinson_samples=None, mle_sigma_thres=0.2, ortho_tf_type=None):
        super().__init__()
        self.operator = operator
        self.y, self.y_flatten = measurement
        self._y1, self._yf1 = self.y, self.y_flatten          # as given (the reference only ever sees batch 1)
        self.guidance = guidance
        self.zeta = zeta
        self.lambda_ = lambda_
        self.eta = eta
        self.num_hutchinson_samples = num_hutchinson_samples
        self.mle_sigma_thres = mle_sigma_thres
        self.device = device
        self.ortho_tf_type = ortho_tf_type
        self.ortho_tf = OrthoTransform(ortho_tf_type)
        self.mat_solver = __MAT_SOLVER__[operator.name]          # KeyError for operators without a solver
        self.lib = L.load()

    @abstractmethod
    def uncond_pred(self, x, sigma):
        raise NotImplementedError

    @abstractmethod
    def _vjp_x0(self, ghat):
        """(d x0_mean / d x)^T ghat for the x of the last uncond_pred."""
        raise NotImplementedError

    # ----------------------------------------------------------------- small kernels ----
    def _combine(self, x0_mean, score, coef):
        """clamp(x0_mean + coef * score, -1, 1) (the final clip of forward() fused in)."""
        hat = torch.empty_like(x0_mean)
        L.check(self.lib.kdip_guidance_combine(L.stream(), L.ptr(x0_mean), L.ptr(score.contiguous()), 1.0, None, 0.0,
                                               float(coef), x0_mean.numel(), L.ptr(hat)))
        return hat

    def _clip(self, x):
        out = torch.empty_like(x)
        L.check(self.lib.kdip_clamp(L.stream(), L.ptr(x.contiguous()), x.numel(), L.ptr(out)))
        return out

    def _solve(self, x0_mean, x0_var, theta0_var):
        var = x0_var if self.ortho_tf_type is None else theta0_var
        return self.mat_solver(self.operator, self.y, x0_mean, var, self.ortho_tf)

    # -------------------------------------------------------------------- dispatch ----
    def _bind_batch(self, B):
        """A call handles B independent batch-1 problems: a batch-1 measurement is shared by all of them
        (B posterior samples of one measurement, the reference's `-n`); otherwise batches must agree."""
        if self.y.shape[0] == B:
            return
        if self._y1.shape[0] == B:
            self.y, self.y_flatten = self._y1, self._yf1
        elif self._y1.shape[0] == 1:
            self.y = self._y1.expand(B, *self._y1.shape[1:]).contiguous()
            self.y_flatten = self._yf1.expand(B, *self._yf1.shape[1:]).contiguous() if self._yf1 is not None else None
        else:
            raise ValueError(f"measurement batch {self._y1.shape[0]} does not match the sample batch {B}")

    def forward(self, x, sigma):
        L.require_gpu()
        x = x.detach().contiguous()
        self._bind_batch(x.shape[0])
        g = self.guidance
        low = sigma_host(sigma) < self.mle_sigma_thres
        if g == "uncond":
            hat_x0 = self._clip(self.uncond_pred(x, sigma)[0])
        elif g == "autoI":
            hat_x0 = self._auto_type_I_guidance_impl(x, sigma)
        elif g == "I":
            hat_x0 = self._type_I_guidance_impl(x, sigma)
        elif g == "II":
            hat_x0 = self._type_II_guidance_impl(x, sigma)
        elif g == "dps":
            hat_x0 = self._dps_guidance_impl(x, sigma)
        elif g == "pgdm":
            hat_x0 = self._pgdm_guidance_impl(x, sigma)
        elif g == "diffpir":
            hat_x0 = self._diffpir_guidance_impl(x, sigma)
        elif g == "stsl":
            hat_x0 = self._stsl_guidance_impl(x, sigma)
        elif g == "dps+mle":
            hat_x0 = self._type_I_guidance_impl(x, sigma) if low else self._dps_guidance_impl(x, sigma)
        elif g == "pgdm+mle":
            hat_x0 = self._type_I_guidance_impl(x, sigma) if low else self._pgdm_guidance_impl(x, sigma)
        elif g == "stsl+mle":
            hat_x0 = self._type_I_guidance_impl(x, sigma) if low else self._stsl_guidance_impl(x, sigma)
        else:
            raise ValueError(f"Invalid guidance type: '{self.guidance}'.")
        return hat_x0.detach()

    # ------------------------------------------------------------------- guidance ----
    def _auto_type_I_guidance_impl(self, x, sigma):
        """grad_x log N(y; A x0, sigma_s^2 I + A C A^T): the gradient reaches x only through the
        mean, i.e. Type-I with the system solved by CG/closed form instead of
        GPyTorch's linear_operator CG (parity unpinned at that third-party boundary)."""
        return self._type_I_guidance_impl(x, sigma)

    def _type_I_guidance_impl(self, x, sigma):
        x0_mean, x0_var, theta0_var = self.uncond_pred(x, sigma)
        mat = self._solve(x0_mean, x0_var, theta0_var)
        likelihood_score = self._vjp_x0(mat)
        return self._combine(x0_mean, likelihood_score, sigma_host(sigma) ** 2)

    def _type_II_guidance_impl(self, x, sigma):
        x0_mean, x0_var, theta0_var = self.uncond_pred(x, sigma)
        mat = self._solve(x0_mean, x0_var, theta0_var)
        var = x0_var if self.ortho_tf_type is None else theta0_var
        ot, iot = self.ortho_tf, self.ortho_tf.inv
        w = ot(mat)
        if var.numel() == 1:
            upd = iot(w) if self.ortho_tf_type is not None else w
            return self._combine(x0_mean, upd, float(var))
        prod = torch.empty_like(w)
        L.check(self.lib.kdip_mul(L.stream(), L.ptr(w.contiguous()), L.ptr(var.contiguous()), w.numel(), L.ptr(prod)))
        return self._combine(x0_mean, iot(prod), 1.0)

    def _dps_guidance_impl(self, x, sigma):
        assert self.zeta is not None, "zeta must be specified for DPS guidance"
        x0_mean = self.uncond_pred(x, sigma)[0]
        # -zeta * grad ||y - A x0||_2  =  (dx0/dx)^T [ zeta * A^T r / ||r|| ], norm per sample
        ghat = self._dps_cotangent(x0_mean)
        likelihood_score = self._vjp_x0(ghat)
        return self._combine(x0_mean, likelihood_score, sigma_host(sigma) ** 2)

    def _pgdm_guidance_impl(self, x, sigma):
        x0_mean = self.uncond_pred(x, sigma)[0]
        s = torch.tensor(sigma_host(sigma), dtype=torch.float32)
        x0_var = (s.pow(2) / (1 + s.pow(2))).reshape(1)
        mat = self.mat_solver(self.operator, self.y, x0_mean, x0_var)
        likelihood_score = self._vjp_x0(mat)
        return self._combine(x0_mean, likelihood_score, float(s.pow(2) * x0_var))

    def _diffpir_guidance_impl(self, x, sigma):
        assert self.lambda_ is not None, "lambda_ must be specified for DiffPIR guidance"
        x0_mean = self.uncond_pred(x, sigma)[0]
        s = torch.tensor(sigma_host(sigma), dtype=torch.float32)
        x0_var = (s.pow(2) / self.lambda_).reshape(1)
        mat = self.mat_solver(self.operator, self.y, x0_mean, x0_var)
        return self._combine(x0_mean, mat, float(x0_var))

    def _axpby(self, x, a, y, b):
        out = torch.empty_like(x)
        L.check(self.lib.kdip_axpby(L.stream(), L.ptr(x.contiguous()), float(a), L.ptr(y.contiguous()) if y is not None else None,
                                    float(b), x.numel(), L.ptr(out)))
        return out

    def _dps_cotangent(self, x0_mean):
        """zeta * A^T r / ||r||_2 with r = y - A x0 (per-sample norm): the cotangent whose VJP is
        -zeta * grad_x ||y - A x0(x)||."""
        ax = self.operator.forward(x0_mean, noiseless=True)
        diff = self._axpby(self.y, 1.0, ax, -1.0)
        atr = self.operator.forward_adjoint(diff).contiguous()
        B = x0_mean.shape[0]
        ghat = torch.empty_like(atr)
        nrm = torch.empty(B, device=x0_mean.device)
        tmp = torch.empty(B, device=x0_mean.device, dtype=torch.float64)
        L.check(self.lib.kdip_dps_normalize(L.stream(), L.ptr(atr), atr[0].numel(), L.ptr(diff), diff[0].numel(),
                                            float(self.zeta), B, L.ptr(ghat), L.ptr(nrm), L.ptr(tmp)))
        return ghat

    def _stsl_guidance_impl(self, x, sigma):
        """STSL (condition.py:185-208): first-order DPS term + Hutchinson estimate of the second-order
        term.  With c = eta * sigma^2 / (n * numel):
            score = J(x)^T [zeta A^T r/||r|| + c sum_k eps_k] - c sum_k J(x + eps_k)^T eps_k
        i.e. one VJP at x and one forward + VJP per probe (no autograd graph is kept)."""
        assert self.zeta is not None and self.eta is not None and self.num_hutchinson_samples is not None, \
            "zeta, eta, and num_hutchinson_samples must be specified for STSL guidance"
        n = self.num_hutchinson_samples
        s2 = sigma_host(sigma) ** 2
        x0_mean = self.uncond_pred(x, sigma)[0]
        cot = self._dps_cotangent(x0_mean)
        probes = getattr(self, "stsl_eps", None)
        eps_list = [torch.randn_like(x) for _ in range(n)] if probes is None else [e.to(x.device) for e in probes]
        c = float(self.eta) * s2 / (n * x[0].numel())
        for eps in eps_list:
            cot = self._axpby(cot, 1.0, eps, c)
        score = self._vjp_x0(cot)
        for eps in eps_list:
            self.uncond_pred(self._axpby(x, 1.0, eps, 1.0), sigma)
            score = self._axpby(score, 1.0, self._vjp_x0(eps), -c)
        return self._combine(x0_mean, score, s2)


class ConditionOpenAIDenoiser(ConditionDenoiser):

    def __init__(self, inner_model, diffusion, x0_cov_type, recon_mse, **kwargs):
        super().__init__(**kwargs)
        self.inner_model = inner_model
        self.diffusion = diffusion
        self.denoiser = OpenAIDenoiser(inner_model, diffusion, device=self.device)
        self.x0_cov_type = x0_cov_type
        self.recon_mse = recon_mse
        if recon_mse is not None:
            # host copies: the analytic lookup is a scalar table read
            self._mse_sigmas = recon_mse['sigmas'].detach().to("cpu", torch.float32)
            self._mse_list = recon_mse['mse_list'].detach().to("cpu", torch.float32)
        self._stash = None

    def uncond_pred(self, x, sigma):
        D = self.diffusion
        s = torch.tensor(sigma_host(sigma), dtype=torch.float32)
        c_in = float(1 / (s ** 2 + 1) ** 0.5)                              # external.py:97-100
        t_frac = self.denoiser.sigma_to_t(sigma)
        t = int(t_frac.reshape(-1)[0].long())                               # floor (condition.py:233)
        B, HW = x.shape[0], x.shape[-1] * x.shape[-2]
        tvec = torch.full((B,), float(t), device=x.device)
        out, _, _ = self.inner_model.forward_raw(x, tvec, in_scale=c_in)
        ct = self.x0_cov_type
        if ct not in ('convert', 'analytic', 'pgdm', 'dps', 'diffpir', 'tmpd'):
            raise ValueError('Invalid posterior covariance type.')
        low = float(s) < self.mle_sigma_thres
        want_var = ct == 'convert' and low
        tables = (c_in, D.f32('sqrt_recip_alphas_cumprod', t), D.f32('sqrt_recipm1_alphas_cumprod', t),
                  D.f32('log_betas', t), D.f32('posterior_log_variance_clipped', t), D.f32('posterior_variance', t),
                  D.f32('posterior_mean_coef1', t))
        import ctypes as C
        t7 = (C.c_float * 7)(*tables)
        x0_mean = torch.empty_like(x)
        x0_raw = torch.empty_like(x)
        var = torch.empty_like(x) if want_var else None
        L.check(self.lib.kdip_x0_epilogue_v1(L.stream(), L.ptr(out), L.ptr(x), B, HW, t7, L.ptr(x0_mean), L.ptr(x0_raw),
                                             L.ptr(var)))
        self._stash = (x0_raw, c_in, tables[1], tables[2], B, HW)
        base = (s.pow(2) / (1 + s.pow(2))).reshape(1)
        if ct == 'convert':
            x0_var = var if low else base                                   # Eq. (22), condition.py:241-248
        elif ct == 'analytic':
            assert self.recon_mse is not None
            if low:
                idx = (self._mse_sigmas - s).abs().argmin()
                x0_var = self._mse_list[idx].reshape(1)
            else:
                x0_var = base
        elif ct == 'pgdm':
            x0_var = base
        elif ct == 'dps':
            x0_var = torch.zeros(1)
        elif ct == 'diffpir':
            assert self.lambda_ is not None
            x0_var = (s.pow(2) / self.lambda_).reshape(1)
        else:  # tmpd: sigma^2 * grad_x sum(x0_mean) = sigma^2 * (dx0/dx)^T 1  (condition.py:268-269)
            ones = torch.ones_like(x)
            x0_var = self._axpby(self._vjp_x0(ones), float(s.pow(2)), None, 0.0)
        return x0_mean, x0_var, x0_var

    # ---- one C entry point per guided call (kdip_guided_call_v1, SURVEY.md 8b `kdip_guided_step`): the same kernels in the same
    # order as uncond_pred -> _solve -> _vjp_x0 -> _combine below, without the ~10 Python / ctypes round trips and the torch.empty
    # temporaries of that path (one cached workspace per batch size).  Taken for every scalar / learned-variance covariance type
    # in the pixel basis; `tmpd` (an extra VJP for the variance) and subclasses with their own uncond_pred keep the stepwise path.
    fused_call = True

    def _type_I_guidance_impl(self, x, sigma):
        ct = self.x0_cov_type
        # (the fused entry point runs the library's own mat-solver: a solver registered through register_mat_solver, or a
        #  mat_solver attribute set by the caller, keeps the stepwise path that goes through self.mat_solver)
        if not (self.fused_call and type(self) is ConditionOpenAIDenoiser and self.ortho_tf_type is None and
                ct in ('convert', 'analytic', 'pgdm', 'dps', 'diffpir') and hasattr(self.operator, "_h") and
                getattr(self.mat_solver, "_kdip_builtin", False)):
            return super()._type_I_guidance_impl(x, sigma)
        S_model = self.inner_model.image_size
        if tuple(x.shape[1:]) != (3, S_model, S_model):
            raise ValueError(f"expected x of shape [B, 3, {S_model}, {S_model}] for this UNet, got {tuple(x.shape)}")
        import ctypes as C
        D = self.diffusion
        s = torch.tensor(sigma_host(sigma), dtype=torch.float32)
        c_in = float(1 / (s ** 2 + 1) ** 0.5)
        t = int(self.denoiser.sigma_to_t(sigma).reshape(-1)[0].long())
        low = float(s) < self.mle_sigma_thres
        base = float(s.pow(2) / (1 + s.pow(2)))
        tensor_var = ct == 'convert' and low
        if ct == 'analytic' and low:
            assert self.recon_mse is not None
            v = float(self._mse_list[(self._mse_sigmas - s).abs().argmin()])
        elif ct == 'dps':
            v = 0.0
        elif ct == 'diffpir':
            assert self.lambda_ is not None
            v = float(s.pow(2) / self.lambda_)
        else:
            v = base
        B, S = x.shape[0], x.shape[-1]
        t7 = (C.c_float * 7)(c_in, D.f32('sqrt_recip_alphas_cumprod', t), D.f32('sqrt_recipm1_alphas_cumprod', t), D.f32('log_betas', t),
                             D.f32('posterior_log_variance_clipped', t), D.f32('posterior_variance', t), D.f32('posterior_mean_coef1', t))
        key = (B, S, x.device)
        if getattr(self, "_fused_ws_key", None) != key:
            self._fused_ws = torch.empty(int(self.lib.kdip_guided_ws_floats(B, S)), device=x.device)
            self._fused_t = torch.empty(B, device=x.device)
            off = (C.c_long * L.GWS_COUNT)()
            L.check(self.lib.kdip_guided_ws_layout(B, S, off, L.GWS_COUNT))
            self._fused_off = list(off)
            self._fused_ws_key = key
        self._fused_t.fill_(float(t))
        op = self.operator
        op._set_ortho(L.OT_NONE)
        y = op._check(self.y)
        hat = torch.empty_like(x)
        iters, info = (C.c_int * B)(), (C.c_int * B)()
        # (dtype "f16x3": the whole call is redone bf16-headed when an operand of one of its convs left the fp16 window -- UNetModel.guarded)
        self.inner_model.before_forward()
        self.inner_model.guarded(lambda: L.check(self.lib.kdip_guided_call_v1(self.inner_model._h, op._h, L.stream(), L.ptr(x), L.ptr(self._fused_t), L.ptr(y), B, t7,
                                                                              float(s), v, int(tensor_var), L.ptr(self._fused_ws), L.ptr(hat), iters, info)))
        op.cg_iters, op.cg_info = list(iters), list(info)
        if any(i > 0 for i in op.cg_info):
            warn('CG not converge.')
        # same stash as uncond_pred leaves behind (x0_raw lives in the call's workspace: valid until the next fused call)
        n3, o = 3 * B * S * S, self._fused_off[L.GWS_X0_RAW]
        self._stash = (self._fused_ws[o:o + n3].view(B, 3, S, S), c_in, t7[1], t7[2], B, S * S)
        return hat

    def _vjp_x0(self, ghat):
        x0_raw, c_in, a_t, b_t, B, HW = self._stash
        ghat = ghat.contiguous()
        cot = torch.empty(B, 6, *ghat.shape[-2:], device=ghat.device)
        g_raw = torch.empty_like(ghat)
        L.check(self.lib.kdip_vjp_cotangent_v1(L.stream(), L.ptr(ghat), L.ptr(x0_raw), B, HW, float(b_t), L.ptr(cot), L.ptr(g_raw)))
        ug = self.inner_model.vjp(cot)
        out = torch.empty_like(ghat)
        L.check(self.lib.kdip_axpby(L.stream(), L.ptr(g_raw), c_in * a_t, L.ptr(ug), c_in, ghat.numel(), L.ptr(out)))
        return out


class ConditionOpenAIDenoiserV2(ConditionDenoiser):

    def __init__(self, denoiser: OpenAIDenoiserV2, **kwargs):
        super().__init__(**kwargs)
        self.denoiser = denoiser
        ortho_tf_type = kwargs.get('ortho_tf_type', None)
        if ortho_tf_type is not None:
            assert ortho_tf_type == denoiser.ortho_tf_type, "ortho_tf_type must match the one used in the denoiser"
        self._stash = None

    def uncond_pred(self, x, sigma):
        s = sigma_host(sigma)
        s32 = torch.tensor(s, dtype=torch.float32)
        c_in = float(1 / (s32 ** 2 + 1) ** 0.5)
        t = self.denoiser.sigma_to_t(sigma)                                  # fractional
        B, HW = x.shape[0], x.shape[-1] * x.shape[-2]
        tvec = torch.full((B,), float(t.reshape(-1)[0]), device=x.device)
        out, cov, _ = self.denoiser.inner_model.forward_raw(x, tvec, in_scale=c_in, want_cov=True)
        low = s < self.mle_sigma_thres
        x0_mean = torch.empty_like(x)
        x0_var = torch.empty_like(x) if low else None
        theta0_var = torch.empty_like(x) if low else None
        L.check(self.lib.kdip_x0_epilogue_v2(L.stream(), L.ptr(out), L.ptr(cov), L.ptr(x), B, HW, s, int(low),
                                             L.ptr(x0_mean), L.ptr(x0_var), L.ptr(theta0_var)))
        self._stash = (s, c_in, B, HW)
        if not low:
            x0_var = theta0_var = (s32.pow(2) / (1 + s32.pow(2))).reshape(1)
        return x0_mean, x0_var, theta0_var

    def _vjp_x0(self, ghat):
        s, c_in, B, HW = self._stash
        ghat = ghat.contiguous()
        cot = torch.empty(B, 6, *ghat.shape[-2:], device=ghat.device)
        L.check(self.lib.kdip_vjp_cotangent_v2(L.stream(), L.ptr(ghat), B, HW, L.ptr(cot)))
        ug = self.denoiser.inner_model.vjp(cot)
        out = torch.empty_like(ghat)
        L.check(self.lib.kdip_axpby(L.stream(), L.ptr(ghat), 1.0, L.ptr(ug), -s * c_in, ghat.numel(), L.ptr(out)))
        return out


# ---------------------------------------------
# Implementation of mat solver (computing v)
# ---------------------------------------------

__MAT_SOLVER__ = {}


def register_mat_solver(name):
    def wrapper(func):
        __MAT_SOLVER__[name] = func
        return func
    return wrapper


def _device_solve(operator, y, x0_mean, theta0_var, ortho_tf):
    mat = operator.solve(y, x0_mean, theta0_var, ortho_tf.code)
    if any(i > 0 for i in operator.cg_info):          # (-1: fixed-trip mode, checked through operator.cg_unconverged())
        warn('CG not converge.')
    return mat


@register_mat_solver('inpainting')
@torch.no_grad()
def inpainting_mat(operator, y, x0_mean, theta0_var, ortho_tf=OrthoTransform()):
    return _device_solve(operator, y, x0_mean, theta0_var, ortho_tf)


@register_mat_solver('gaussian_blur')
@torch.no_grad()
def gaussian_blur_mat(operator, y, x0_mean, theta0_var, ortho_tf=OrthoTransform()):
    return _device_solve(operator, y, x0_mean, theta0_var, ortho_tf)


@register_mat_solver('motion_blur')
@torch.no_grad()
def motion_blur_mat(operator, y, x0_mean, theta0_var, ortho_tf=OrthoTransform()):
    return _device_solve(operator, y, x0_mean, theta0_var, ortho_tf)


@register_mat_solver('super_resolution')
@torch.no_grad()
def super_resolution_mat(operator, y, x0_mean, theta0_var, ortho_tf=OrthoTransform()):
    return _device_solve(operator, y, x0_mean, theta0_var, ortho_tf)


for _f in (inpainting_mat, gaussian_blur_mat, motion_blur_mat, super_resolution_mat):
    _f._kdip_builtin = True          # solved by the library's own kdip_op_solve: what kdip_guided_call_v1 runs internally
