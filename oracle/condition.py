"""Oracle: guided denoiser (condition/condition.py:44-300) on torch-CPU fp32.

TEST INFRASTRUCTURE (see oracle/__init__.py).

Batch semantics (the reference asserts batch 1, condition.py:84): B independent
batch-1 problems sharing one sigma per call; norms (DPS) and CG are per sample.
"""
import torch
from torch.autograd import grad

from .tables import DiffusionTables
from .unet import unet_forward
from .solvers import MAT_SOLVER
from .transforms import OrthoTransform


def _bc(v, x):
    """append dims for broadcasting a [B] or [1] tensor against x."""
    return v.view(-1, *([1] * (x.ndim - 1)))


class GuidedDenoiser:
    """ConditionOpenAIDenoiser (v1, condition.py:211-274) and
    ConditionOpenAIDenoiserV2 (v2=True, condition.py:277-300) in one class."""

    def __init__(self, sd, cfg, operator, measurement, guidance, x0_cov_type="convert",
                 recon_mse=None, zeta=None, lambda_=None, mle_sigma_thres=0.2,
                 ortho_tf_type=None, v2=False, tables=None, eta=None, num_hutchinson_samples=None):
        self.sd, self.cfg = sd, cfg
        self.operator = operator
        self.y, self.y_flatten = measurement
        self.guidance = guidance
        self.x0_cov_type = x0_cov_type
        self.recon_mse = recon_mse
        self.zeta, self.lambda_ = zeta, lambda_
        self.eta, self.num_hutchinson_samples = eta, num_hutchinson_samples
        self.mle_sigma_thres = mle_sigma_thres
        self.ortho_tf_type = ortho_tf_type
        self.ortho_tf = OrthoTransform(ortho_tf_type)
        self.mat_solver = MAT_SOLVER[operator.name]     # KeyError for unknown (condition.py:71)
        self.v2 = v2
        self.D = tables or DiffusionTables()
        self.cg_stats = {}
        # The guided output is DISCONTINUOUS in the UNet output where |x0_raw| crosses 1 (the VJP goes through the clamp,
        # condition.py:231): a pixel whose |x0_raw| equals 1 to within rounding has two correct answers that differ by O(1) over its
        # receptive field.  The oracle owns the set of such pixels: `last_borderline` lists the indices with | |x0_raw| - 1 | <
        # clamp_borderline_tol in ITS OWN x0_raw.  `clamp_flip` (a list of index tuples) asks for the answer with the clamp-gradient
        # mask flipped at those pixels; a listed pixel that is not borderline by the oracle's own numbers is refused.  Nothing else
        # of a run under test can reach the oracle.
        self.clamp_flip = None
        self.clamp_borderline_tol = 1e-4
        self.last_x0_raw = None
        self.last_borderline = []

    # ------------------------------------------------------------- uncond ----
    def uncond_pred(self, x, sigma):
        if self.v2:
            return self._uncond_pred_v2(x, sigma)
        D = self.D
        s0 = sigma[:1]
        if self.x0_cov_type == "tmpd" and not x.requires_grad:
            x = x.requires_grad_()                                # condition.py:236-237
        c_in = 1 / (s0 ** 2 + 1) ** 0.5                          # external.py:97-100
        t = D.sigma_to_t(sigma).long()                            # condition.py:233 (floor)
        x_in = x * c_in
        out = unet_forward(self.sd, self.cfg, x_in, t)            # respace.py:123-128 identity map
        eps, v = torch.split(out, 3, dim=1)                       # gaussian_diffusion.py:262-264
        min_log = _bc(D.f32(D.posterior_log_variance_clipped, t), x)
        max_log = _bc(D.f32(D.log_betas, t), x)
        frac = (v + 1) / 2
        variance = torch.exp(frac * max_log + (1 - frac) * min_log)                       # :270-276
        x0_raw = (_bc(D.f32(D.sqrt_recip_alphas_cumprod, t), x) * x_in
                  - _bc(D.f32(D.sqrt_recipm1_alphas_cumprod, t), x) * eps)
        self.last_x0_raw = x0_raw.detach()
        x0_mean = x0_raw.clamp(-1, 1)                                                      # :293-311,328-333
        near = (x0_raw.detach().abs() - 1).abs() < self.clamp_borderline_tol
        self.last_borderline = [tuple(i) for i in near.nonzero().tolist()]
        if self.clamp_flip:
            mask = x0_raw.detach().abs() <= 1
            for idx in self.clamp_flip:
                idx = tuple(int(v) for v in idx)
                if not bool(near[idx]):
                    raise ValueError(f"clamp_flip: pixel {idx} is not borderline in the oracle's own x0_raw ({float(x0_raw.detach()[idx])!r})")
                mask[idx] = ~mask[idx]
            x0_mean = torch.where(mask, x0_raw, x0_mean.detach())      # same values, the flipped gradient mask
        ct = self.x0_cov_type
        base = s0.pow(2) / (1 + s0.pow(2))
        if ct == "convert":
            if float(s0) < self.mle_sigma_thres:
                x0_var = ((variance - _bc(D.f32(D.posterior_variance, t), x))
                          / _bc(D.f32(D.posterior_mean_coef1, t), x).pow(2)).clip(min=1e-6)   # Eq. 22, :242-246
            else:
                x0_var = base
        elif ct == "analytic":
            if float(s0) < self.mle_sigma_thres:
                idx = (self.recon_mse["sigmas"] - s0[0]).abs().argmin()        # :253-254
                x0_var = self.recon_mse["mse_list"][idx].reshape(1)
            else:
                x0_var = base
        elif ct == "pgdm":
            x0_var = base
        elif ct == "dps":
            x0_var = torch.zeros(1)
        elif ct == "diffpir":
            x0_var = s0.pow(2) / self.lambda_
        elif ct == "tmpd":
            # condition.py:268-269: sigma^2 * grad_x sum(x0_mean) (an extra VJP with an all-ones cotangent)
            x0_var = grad(x0_mean.sum(), x, retain_graph=True)[0] * s0.pow(2)
        else:
            raise ValueError("Invalid posterior covariance type.")
        return x0_mean, x0_var, x0_var

    def _uncond_pred_v2(self, x, sigma):
        """condition.py:287-300 + OpenAIDenoiserV2.forward (external.py:161-169):
        fractional t, no clamp, variances from the 1x1 `out_cov` head on feature h."""
        import torch.nn.functional as F
        s0 = sigma[:1]
        c_in = 1 / (s0 ** 2 + 1) ** 0.5
        c_out = -s0
        t = self.D.sigma_to_t(sigma)
        out, feat = unet_forward(self.sd, self.cfg, x * c_in, t, return_feature=True)
        eps = out.chunk(2, dim=1)[0]
        logvar, logvar_ot = F.conv2d(feat, self.sd["out_cov.weight"], self.sd["out_cov.bias"]).chunk(2, dim=1)
        x0_mean = eps * c_out + x
        if float(s0) < self.mle_sigma_thres:
            x0_var = logvar.exp() * c_out.pow(2)
            theta0_var = logvar_ot.exp() * c_out.pow(2)
        else:
            x0_var = theta0_var = s0.pow(2) / (1 + s0.pow(2))
        return x0_mean, x0_var, theta0_var

    # ----------------------------------------------------------- guidance ----
    def _solve(self, x0_mean, x0_var, theta0_var):
        var = x0_var if self.ortho_tf_type is None else theta0_var
        return self.mat_solver(self.operator, self.y, x0_mean.detach(), var.detach(),
                               self.ortho_tf, cg_stats=self.cg_stats)

    def _type_I(self, x, sigma):
        """condition.py:167-174. autoI (:133-138) has the same gradient."""
        x = x.detach().requires_grad_()
        x0_mean, x0_var, theta0_var = self.uncond_pred(x, sigma)
        mat = self._solve(x0_mean, x0_var, theta0_var)
        score = grad((mat.detach() * x0_mean).sum(), x)[0]
        return x0_mean + sigma[:1].pow(2) * score

    def _type_II(self, x, sigma):
        """condition.py:176-183."""
        import contextlib
        ctx = contextlib.nullcontext() if self.x0_cov_type == "tmpd" else torch.no_grad()
        with ctx:
            x0_mean, x0_var, theta0_var = self.uncond_pred(x.detach(), sigma)
            mat = self._solve(x0_mean, x0_var, theta0_var)
            var = x0_var if self.ortho_tf_type is None else theta0_var
            return (x0_mean + self.ortho_tf.inv(self.ortho_tf(mat) * var)).detach()

    def _dps(self, x, sigma):
        """condition.py:140-148; the 2-norm is per sample in the batched form."""
        assert self.zeta is not None, "zeta must be specified for DPS guidance"
        x = x.detach().requires_grad_()
        x0_mean = self.uncond_pred(x, sigma)[0]
        diff = self.y - self.operator.forward(x0_mean, noiseless=True)
        norm = diff.flatten(1).norm(dim=1).sum()
        score = -grad(norm, x)[0] * self.zeta
        return x0_mean + sigma[:1].pow(2) * score

    def _pgdm(self, x, sigma):
        """condition.py:150-157."""
        x = x.detach().requires_grad_()
        x0_mean = self.uncond_pred(x, sigma)[0]
        s0 = sigma[:1]
        x0_var = s0.pow(2) / (1 + s0.pow(2))
        mat = self.mat_solver(self.operator, self.y, x0_mean.detach(), x0_var)
        score = grad((mat.detach() * x0_mean).sum(), x)[0] * x0_var
        return x0_mean + s0.pow(2) * score

    def _diffpir(self, x, sigma):
        """condition.py:159-165."""
        assert self.lambda_ is not None, "lambda_ must be specified for DiffPIR guidance"
        with torch.no_grad():
            x0_mean = self.uncond_pred(x, sigma)[0]
            x0_var = sigma[:1].pow(2) / self.lambda_
            mat = self.mat_solver(self.operator, self.y, x0_mean, x0_var)
            return x0_mean + mat * x0_var

    def _stsl(self, x, sigma, eps_list=None):
        """condition.py:185-208; per-sample norm and per-sample numel in the batched form.
        eps_list (optional) supplies the Hutchinson probes instead of torch.randn_like."""
        assert self.zeta is not None and self.eta is not None and self.num_hutchinson_samples is not None, \
            "zeta, eta, and num_hutchinson_samples must be specified for STSL guidance"
        x = x.detach().requires_grad_()
        x0_mean = self.uncond_pred(x, sigma)[0]
        diff = self.y - self.operator.forward(x0_mean, noiseless=True)
        first = -diff.flatten(1).norm(dim=1).sum()
        second = 0
        for k in range(self.num_hutchinson_samples):
            eps = torch.randn_like(x) if eps_list is None else eps_list[k]
            inc = self.uncond_pred(x + eps, sigma)[0]
            second = second + -((inc - x0_mean) * eps).sum() * sigma[:1].pow(2)
        second = second / self.num_hutchinson_samples
        loss = self.zeta * first + (self.eta / x[0].numel()) * second
        score = grad(loss.sum(), x)[0]
        return x0_mean + sigma[:1].pow(2) * score

    def __call__(self, x, sigma):
        """ConditionDenoiser.forward (condition.py:83-131)."""
        g = self.guidance
        low = float(sigma[0]) < self.mle_sigma_thres
        if g == "uncond":
            with torch.no_grad():
                hat = self.uncond_pred(x, sigma)[0]
        elif g in ("I", "autoI"):
            hat = self._type_I(x, sigma)
        elif g == "II":
            hat = self._type_II(x, sigma)
        elif g == "dps":
            hat = self._dps(x, sigma)
        elif g == "pgdm":
            hat = self._pgdm(x, sigma)
        elif g == "diffpir":
            hat = self._diffpir(x, sigma)
        elif g == "stsl":
            hat = self._stsl(x, sigma, getattr(self, "stsl_eps", None))
        elif g == "stsl+mle":
            hat = self._type_I(x, sigma) if low else self._stsl(x, sigma, getattr(self, "stsl_eps", None))
        elif g == "dps+mle":
            hat = self._type_I(x, sigma) if low else self._dps(x, sigma)
        elif g == "pgdm+mle":
            hat = self._type_I(x, sigma) if low else self._pgdm(x, sigma)
        else:
            raise ValueError(f"Invalid guidance type: '{g}'.")
        return hat.clip(-1, 1).detach()
