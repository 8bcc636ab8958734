"""Oracle: mat-solvers  mat = A^T (sigma_s^2 I + A C A^T)^-1 (y - A x0)
(condition/condition.py:307-439) on torch-CPU fp32.

TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference's tensor-variance branch calls scipy.sparse.linalg.cg(tol=1e-4,
maxiter=1000) on host float32; `cg_batched` restates that algorithm (legacy
criterion ||r|| <= tol*||b||, x0 = 0, no preconditioner) with every batch
sample an independent solve that freezes when it converges.
"""
import torch
from torch.fft import fft2, ifft2

from .operators import splits, upsample_zero, downsample
from .transforms import OrthoTransform


def cg_batched(matvec, b, tol=1e-4, maxiter=1000):
    """Per-sample CG. b: [B, ...]. Returns (x, iters[B], info[B])."""
    B = b.shape[0]
    dims = tuple(range(1, b.ndim))
    shp = (B,) + (1,) * (b.ndim - 1)

    def dot(u, v):
        return (u * v).sum(dim=dims)

    x = torch.zeros_like(b)
    r = b.clone()
    p = torch.zeros_like(b)
    atol = tol * dot(b, b).sqrt()
    rho_prev = torch.ones(B, dtype=b.dtype)
    iters = torch.zeros(B, dtype=torch.long)
    active = torch.ones(B, dtype=torch.bool)
    for it in range(maxiter):
        rn = dot(r, r).sqrt()
        active = active & ~(rn < atol) & (atol > 0)
        if not active.any():
            break
        rho = dot(r, r)
        beta = torch.where(active, rho / rho_prev, torch.zeros_like(rho)) if it > 0 else torch.zeros_like(rho)
        p = torch.where(active.view(shp), r + beta.view(shp) * p, p)
        q = matvec(p)
        alpha = torch.where(active, rho / dot(p, q), torch.zeros_like(rho))
        x = x + alpha.view(shp) * p
        r = r - alpha.view(shp) * q
        rho_prev = torch.where(active, rho, rho_prev)
        iters = iters + active.long()
    info = torch.where(active, torch.full_like(iters, maxiter), torch.zeros_like(iters))
    return x, iters, info


def _is_scalar(v):
    return v.numel() == 1


@torch.no_grad()
def inpainting_mat(operator, y, x0_mean, theta0_var, ortho_tf=OrthoTransform(), cg_stats=None):
    """condition/condition.py:317-348."""
    mask = operator.mask
    sigma_s = operator.sigma_s.clip(min=0.001)
    if _is_scalar(theta0_var):
        return (mask * y - mask * x0_mean) / (sigma_s.pow(2) + theta0_var)
    ot, iot = ortho_tf, ortho_tf.inv

    def A(u):
        return sigma_s ** 2 * u + mask * iot(theta0_var * ot(u))
    b = mask * y - mask * x0_mean
    u, iters, info = cg_batched(A, b)
    if cg_stats is not None:
        cg_stats.update(iters=iters, info=info)
    return u


@torch.no_grad()
def deblur_mat(operator, y, x0_mean, theta0_var, ortho_tf=OrthoTransform(), cg_stats=None):
    """condition/condition.py:351-386 (gaussian_blur :389-392, motion_blur :395-398)."""
    sigma_s = operator.sigma_s.clip(min=0.001)
    FB, FBC, F2B, _ = operator.pre_calculated
    if _is_scalar(theta0_var):
        return ifft2(fft2(y - ifft2(FB * fft2(x0_mean))) / (sigma_s.pow(2) + theta0_var * F2B) * FBC).real
    ot, iot = ortho_tf, ortho_tf.inv

    def A(u):
        return sigma_s ** 2 * u + ifft2(FB * fft2(iot(theta0_var * ot(ifft2(FBC * fft2(u)).real)))).real
    b = y - ifft2(FB * fft2(x0_mean)).real
    u, iters, info = cg_batched(A, b)
    if cg_stats is not None:
        cg_stats.update(iters=iters, info=info)
    return ifft2(FBC * fft2(u)).real


@torch.no_grad()
def super_resolution_mat(operator, y, x0_mean, theta0_var, ortho_tf=OrthoTransform(), cg_stats=None):
    """condition/condition.py:401-439."""
    sigma_s = operator.sigma_s.clip(min=0.001).clip(min=1e-2)
    sf = operator.scale_factor
    FB, FBC, F2B, _ = operator.pre_calculated
    if _is_scalar(theta0_var):
        invW = torch.mean(splits(F2B, sf), dim=-1, keepdim=False)
        # note: the inner residual stays complex (no .real before the small FFT, :410)
        return ifft2(FBC * (fft2(y - downsample(ifft2(FB * fft2(x0_mean)), sf))
                            / (sigma_s.pow(2) + theta0_var * invW)).repeat(1, 1, sf, sf)).real
    ot, iot = ortho_tf, ortho_tf.inv

    def A(u):
        v = sigma_s ** 2 * u + downsample(
            ifft2(FB * fft2(iot(theta0_var * ot(ifft2(FBC * fft2(upsample_zero(u, sf))).real)))), sf)
        return v.real
    b = (y - downsample(ifft2(FB * fft2(x0_mean)), sf)).real
    u, iters, info = cg_batched(A, b)
    if cg_stats is not None:
        cg_stats.update(iters=iters, info=info)
    return ifft2(FBC * fft2(upsample_zero(u, sf))).real


MAT_SOLVER = {"inpainting": inpainting_mat, "gaussian_blur": deblur_mat,
              "motion_blur": deblur_mat, "super_resolution": super_resolution_mat}
