"""Measurement operators y = A x + n -- plug-in surface of condition/measurements.py
(registry :26-39, SuperResolutionOperator :86-122, MotionBlurOperator :125-160,
GaussialBlurOperator :163-199, InpaintingOperator :202-244, MaskGenerator :247-324).

Same names, constructor kwargs (= the YAML keys), attributes (`name, sigma_s, in_shape, mask,
pre_calculated, scale_factor, get_kernel()`) and forward/transpose signatures; the tensor
work is done by libkdip_hip (circular blur as LDS-staged stencils or the LDS FFT, gather /
scatter at bit-exact indices, antialiased-cubic Resizer).  Batches are B independent problems.
"""
import ctypes as C
import os
from abc import ABC, abstractmethod

import numpy as np
import torch

from . import _lib as L

KERNEL_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kernels")


_NOISE_RNG = "device"


def set_noise_rng(where):
    """'device' (default): measurement noise from the device generator (torch.randn_like on the HIP tensor).  'cpu': drawn from
    torch's CPU generator and copied over -- the random stream of the reference's CPU path, so a harness run can be compared
    value for value with a run of the reference (or of a CPU restatement of it) on the same seeds (sample_condition.py --cpu-rng)."""
    global _NOISE_RNG
    assert where in ("device", "cpu")
    _NOISE_RNG = where


def _randn_like(y):
    if _NOISE_RNG == "cpu":
        return torch.randn(y.shape, dtype=y.dtype).to(y.device)
    return torch.randn_like(y)


__OPERATOR__ = {}


def register_operator(name: str):
    def wrapper(cls):
        if __OPERATOR__.get(name, None):
            raise NameError(f"Name {name} is already registered!")
        cls.name = name
        __OPERATOR__[name] = cls
        return cls
    return wrapper


def get_operator(name: str, **kwargs):
    if __OPERATOR__.get(name, None) is None:
        raise NameError(f"Name {name} is not defined.")
    return __OPERATOR__[name](**kwargs)


def _dev_index(device):
    d = torch.device(device)
    if d.type != "cuda":
        raise L.KdipError(f"kdip_amd operators run on an MI355X; got device '{device}' (no CPU fallback)")
    return d.index if d.index is not None else torch.cuda.current_device()


class LinearOperator(ABC):
    """Base class.  Sub-classes own a `kdip_op` context (device OTF / PSF / mask / CG workspace)."""

    _h = None

    @abstractmethod
    def forward(self, data, flatten=False, noiseless=False):
        raise NotImplementedError("The class {} requires a forward function!".format(self.__class__.__name__))

    def _make_ctx(self, kind, size, sf, sigma_s, device):
        L.require_gpu()
        self.lib = L.load()
        self.device = device
        self._dev_index = _dev_index(device)
        h = C.c_void_p()
        L.check(self.lib.kdip_op_create(self._dev_index, kind, size, sf, float(sigma_s), C.byref(h)))
        self._h = h
        self._ortho_code = L.OT_NONE

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.lib.kdip_op_destroy(h)
            self._h = None

    def _check(self, x):
        if not (x.is_cuda and x.dtype == torch.float32):
            raise L.KdipError("kdip_amd operators need float32 CUDA(HIP) tensors")
        return x.contiguous()

    def _apply(self, x, adjoint, out_shape):
        x = self._check(x)
        out = torch.empty(out_shape, device=x.device, dtype=torch.float32)
        L.check(self.lib.kdip_op_apply(self._h, L.stream(), L.ptr(x), x.shape[0], int(adjoint), L.ptr(out)))
        return out

    def _set_ortho(self, code):
        if code != self._ortho_code:
            L.check(self.lib.kdip_op_set_ortho(self._h, code))
            self._ortho_code = code

    def solve(self, y, x0_mean, theta0_var, ortho_code=L.OT_NONE):
        """mat = A^T (sigma_s^2 I + A C A^T)^-1 (y - A x0): closed form for a 1-element variance,
        batched on-device CG otherwise.  Sets self.cg_iters / self.cg_info (per sample)."""
        y, x0_mean = self._check(y), self._check(x0_mean)
        B = x0_mean.shape[0]
        if y.shape[0] != B:
            raise ValueError(f"solve: measurement batch {y.shape[0]} != estimate batch {B} (B independent problems per call; "
                             "ConditionDenoiser broadcasts a batch-1 measurement)")
        self._set_ortho(ortho_code)
        mat = torch.empty_like(x0_mean)
        iters = (C.c_int * B)()
        info = (C.c_int * B)()
        if theta0_var.numel() == 1:
            L.check(self.lib.kdip_op_solve(self._h, L.stream(), L.ptr(y), L.ptr(x0_mean), float(theta0_var), None, B,
                                           L.ptr(mat), iters, info))
        else:
            vt = self._check(theta0_var.expand_as(x0_mean))
            L.check(self.lib.kdip_op_solve(self._h, L.stream(), L.ptr(y), L.ptr(x0_mean), 0.0, L.ptr(vt), B, L.ptr(mat),
                                           iters, info))
        self.cg_iters, self.cg_info = list(iters), list(info)
        return mat

    def set_cg_fixed_trips(self, trips):
        """trips > 0: every later tensor-variance solve runs exactly `trips` CG iterations with no host read (capture-safe: samples
        freeze themselves on the device when they converge); 0: adaptive (flags read back every second iteration)."""
        L.check(self.lib.kdip_op_set_cg_fixed_trips(self._h, int(trips)))
        self._cg_fixed_trips = int(trips)

    def workspace_generation(self):
        """Bumped whenever the operator context re-allocates its solver workspace (captured hipGraphs hold the old buffers)."""
        return int(self.lib.kdip_op_workspace_generation(self._h))

    def cg_unconverged(self):
        """Number of fixed-trip CG solves since the last query that stopped with an unconverged sample (sticky device counter, read
        and cleared; synchronises the stream)."""
        n = C.c_int(0)
        L.check(self.lib.kdip_op_cg_unconverged(self._h, L.stream(), C.byref(n)))
        return int(n.value)

    def forward_adjoint(self, r):
        """True adjoint of the noiseless `forward` (what autograd applies in DPS, condition.py:143-146)."""
        return self.transpose(r)


def _fft2(lib, x, size, inverse=False, real_out=False):
    """torch.fft.fft2 of real/complex [.., S, S] through the LDS FFT (used for `pre_calculated`)."""
    planes = int(np.prod(x.shape[:-2])) if not x.is_complex() else int(np.prod(x.shape[:-2]))
    src = torch.view_as_real(x).contiguous() if x.is_complex() else x.contiguous()
    out = torch.empty(*x.shape, 2, device=x.device) if not real_out else torch.empty(x.shape, device=x.device)
    tmp = torch.empty(*x.shape, 2, device=x.device)
    L.check(lib.kdip_fft2(L.stream(), size, L.ptr(src), 0 if x.is_complex() else 1, L.ptr(out), int(real_out), planes,
                          int(inverse), L.ptr(tmp)))
    return out if real_out else torch.view_as_complex(out)


class _FFTModelMixin:
    """Blur-type operators expose `pre_calculated = (FB, FBC, F2B, FBFy)` like the reference
    (utils_sisr.pre_calculate, :79-96); it is built lazily -- the solvers use the device
    context, not these tensors."""

    _meas_for_pre = None

    def _otf(self):
        S = self.in_shape[-1]
        otf = torch.empty(S, S, 2, device=self.device)
        L.check(self.lib.kdip_op_get_otf(self._h, L.stream(), L.ptr(otf)))
        return torch.view_as_complex(otf).view(1, 1, S, S)

    @property
    def pre_calculated(self):
        if self._meas_for_pre is None:
            return None
        FB = self._otf()
        FBC = torch.conj(FB)
        F2B = torch.abs(FB) ** 2
        y = self._meas_for_pre
        sf = getattr(self, "scale_factor", 1)
        if sf != 1:
            up = torch.zeros(y.shape[0], y.shape[1], y.shape[2] * sf, y.shape[3] * sf, device=y.device)
            up[..., ::sf, ::sf] = y
            y = up
        return FB, FBC, F2B, FBC * _fft2(self.lib, y, y.shape[-1])


def _load_psf(fname):
    return np.load(os.path.join(KERNEL_DIR, fname))


class _BlurBase(_FFTModelMixin, LinearOperator):
    def _init_blur(self, in_shape, kernel_size, sigma_s, device, psf, separable):
        assert psf.shape == (kernel_size, kernel_size)
        self.kernel_size = kernel_size
        self.in_shape = tuple(in_shape)
        S = self.in_shape[-1]
        self._make_ctx(L.OP_BLUR, S, 1, sigma_s, device)
        self.sigma_s = torch.Tensor([sigma_s]).to(device)
        psf32 = np.ascontiguousarray(psf.astype(np.float32))          # torch.Tensor(...) cast, measurements.py:159,173
        self.kernel = torch.from_numpy(psf32).to(device)
        L.check(self.lib.kdip_op_set_psf(self._h, C.c_void_p(psf32.ctypes.data), kernel_size, kernel_size))
        if separable:
            # Gaussian PSF is rank-1 (sigma_2/sigma_1 = 1.8e-8): row/col factors from the float64 file
            p64 = psf.astype(np.float64)
            kr = np.ascontiguousarray((p64.sum(1) / p64.sum()).astype(np.float32))
            kc = np.ascontiguousarray(p64.sum(0).astype(np.float32))
            L.check(self.lib.kdip_op_set_separable(self._h, C.c_void_p(kr.ctypes.data), C.c_void_p(kc.ctypes.data), kernel_size))

    def forward(self, data, flatten=False, noiseless=False):
        y = self._apply(data, False, data.shape)
        if not noiseless:
            y += self.sigma_s * _randn_like(y)
        self._meas_for_pre = y
        if flatten:
            return y, y.reshape(y.shape[0], -1)
        return y

    def transpose(self, y, flatten=False):
        if flatten:
            y = y.reshape(y.shape[0], *self.in_shape[-3:])
        return self._apply(y, True, y.shape)

    def get_kernel(self):
        return self.kernel.view(1, 1, self.kernel_size, self.kernel_size)


@register_operator(name='motion_blur')
class MotionBlurOperator(_BlurBase):
    def __init__(self, in_shape, kernel_size, intensity, sigma_s, device):
        self._init_blur(in_shape, kernel_size, sigma_s, device, _load_psf('motion_ks61_std0.5.npy'), separable=False)


@register_operator(name='gaussian_blur')
class GaussialBlurOperator(_BlurBase):
    def __init__(self, in_shape, kernel_size, intensity, sigma_s, device):
        self._init_blur(in_shape, kernel_size, sigma_s, device, _load_psf('gaussian_ks61_std3.0.npy'), separable=True)


GaussianBlurOperator = GaussialBlurOperator   # the reference's spelling is kept as the primary name


# ---- antialiased cubic resize tables (host; condition/dps_utils/resizer.py:104-167) -----
def _cubic(x):
    a = np.abs(x)
    return ((1.5 * a ** 3 - 2.5 * a ** 2 + 1) * (a <= 1)
            + (-0.5 * a ** 3 + 2.5 * a ** 2 - 4 * a + 2) * ((1 < a) & (a <= 2)))


def cubic_resize_tables(n_in, n_out, scale):
    """Per-axis (weights fp32 [n_out, taps], source index int32 [n_out, taps]) of the antialiased
    cubic resize with mirrored borders; zero-weight columns trimmed."""
    width = 4.0
    if scale < 1:
        kern = lambda u: scale * _cubic(scale * u)
        width = width / scale
    else:
        kern = _cubic
    centers = (np.arange(1, n_out + 1) - (n_out - n_in * scale) / 2) / scale + 0.5 * (1 - 1 / scale)
    left = np.floor(centers - width / 2)
    ntap = int(np.ceil(width)) + 2
    src = np.int16(left[:, None] + np.arange(ntap) - 1)
    wts = kern(centers[:, None] - src - 1.0)
    tot = wts.sum(1)
    tot[tot == 0] = 1.0
    wts = wts / tot[:, None]
    mirror = np.concatenate((np.arange(n_in), np.arange(n_in - 1, -1, -1)))
    src = mirror[np.mod(src, mirror.shape[0])]
    keep = np.any(wts, axis=0)
    return np.ascontiguousarray(wts[:, keep].astype(np.float32)), np.ascontiguousarray(src[:, keep].astype(np.int32))


def transpose_resize_tables(w, f, n_in):
    """Tables of the ADJOINT of a resize axis as another gather: for every input index `src` the (output o, weight) pairs whose field
    of view contains it, in (o, tap) order, padded to the longest list with weight 0 -- the adjoint then runs on the forward kernel
    (kdip_resize_axis with n_in and n_out exchanged): coalesced, no atomics, a fixed summation order."""
    n_out, taps = w.shape
    lists = [[] for _ in range(n_in)]
    for o in range(n_out):
        for t in range(taps):
            if w[o, t] != 0.0:
                lists[int(f[o, t])].append((o, float(w[o, t])))
    tmax = max(1, max(len(l) for l in lists))
    wt = np.zeros((n_in, tmax), np.float32)
    ft = np.zeros((n_in, tmax), np.int32)
    for src, l in enumerate(lists):
        for j, (o, v) in enumerate(l):
            wt[src, j] = v
            ft[src, j] = o
    return wt, ft


@register_operator(name='super_resolution')
class SuperResolutionOperator(_FFTModelMixin, LinearOperator):
    """forward = antialiased bicubic 1/sf (Resizer); transpose / solvers model A as circular
    bicubic-kernel blur + stride-sf decimation -- the two differ, both reproduced as-is
    (measurements.py:86-122)."""

    def __init__(self, in_shape, scale_factor, sigma_s, device):
        self.in_shape = tuple(in_shape)
        self.scale_factor = scale_factor
        S = self.in_shape[-1]
        self._make_ctx(L.OP_SR, S, scale_factor, sigma_s, device)
        self.sigma_s = torch.Tensor([sigma_s]).to(device)
        k_index = scale_factor - 2 if scale_factor < 5 else 2
        z = np.load(os.path.join(KERNEL_DIR, 'kernels_bicubicx234.npz'))
        k = np.ascontiguousarray(z[f"sf{k_index + 2}"].astype(np.float32))
        self.kernel = torch.from_numpy(k)
        L.check(self.lib.kdip_op_set_psf(self._h, C.c_void_p(k.ctypes.data), k.shape[0], k.shape[1]))
        out_shape = tuple(int(s / scale_factor) for s in in_shape[-2:])
        self.out_shape = (1, 3, *out_shape)
        self._tables, self._tables_T = {}, {}          # forward gather tables / their transposes (adjoint as a gather)
        for n_in in set(self.in_shape[-2:]):
            n_out = int(np.ceil(n_in / scale_factor))
            w, f = cubic_resize_tables(n_in, n_out, 1.0 / scale_factor)
            self._tables[n_in] = (torch.from_numpy(w).to(device), torch.from_numpy(f).to(device), w.shape[1], n_out)
            wt, ft = transpose_resize_tables(w, f, n_in)
            self._tables_T[n_in] = (torch.from_numpy(wt).to(device), torch.from_numpy(ft).to(device), wt.shape[1], n_out)

    def _resize(self, x, adjoint=False):
        """W axis first, then H (np.argsort of equal scales in the reference); adjoint reverses."""
        x = self._check(x)
        B, Cc, H, W = x.shape if not adjoint else (x.shape[0], x.shape[1], self.in_shape[-2], self.in_shape[-1])
        wW, fW, tW, oW = self._tables[W]
        wH, fH, tH, oH = self._tables[H]
        planes = B * Cc
        lib, st = self.lib, L.stream()
        if not adjoint:
            t = torch.empty(B, Cc, H, oW, device=x.device)
            L.check(lib.kdip_resize_axis(st, L.ptr(x), L.ptr(wW), L.ptr(fW), tW, W, oW, H, 1, planes, 0, L.ptr(t)))
            y = torch.empty(B, Cc, oH, oW, device=x.device)
            L.check(lib.kdip_resize_axis(st, L.ptr(t), L.ptr(wH), L.ptr(fH), tH, H, oH, oW, 0, planes, 0, L.ptr(y)))
            return y
        # adjoint: the transposed tables on the forward (gather) kernel, axes in reverse order: [oH, oW] -> [H, oW] -> [H, W]
        wHt, fHt, tHt, _ = self._tables_T[H]
        wWt, fWt, tWt, _ = self._tables_T[W]
        t = torch.empty(B, Cc, H, oW, device=x.device)
        L.check(lib.kdip_resize_axis(st, L.ptr(x), L.ptr(wHt), L.ptr(fHt), tHt, oH, H, oW, 0, planes, 0, L.ptr(t)))
        g = torch.empty(B, Cc, H, W, device=x.device)
        L.check(lib.kdip_resize_axis(st, L.ptr(t), L.ptr(wWt), L.ptr(fWt), tWt, oW, W, H, 1, planes, 0, L.ptr(g)))
        return g

    def forward(self, data, flatten=False, noiseless=False):
        y = self._resize(data)
        if not noiseless:
            y += self.sigma_s * _randn_like(y)
        self._meas_for_pre = y
        if flatten:
            return y, y.reshape(y.shape[0], -1)
        return y

    def transpose(self, y, flatten=False):
        if flatten:
            y = y.reshape(y.shape[0], *self.out_shape[-3:])
        return self._apply(y, True, (y.shape[0], 3, *self.in_shape[-2:]))

    def forward_adjoint(self, r):
        return self._resize(r, adjoint=True)

    def get_kernel(self):
        return self.kernel.view(1, 1, *self.kernel.shape)


@register_operator(name='inpainting')
class InpaintingOperator(LinearOperator):
    '''Pre-defined mask -> masked image; `flatten=True` also returns the kept pixels in
    (c,h,w)-lexicographic order (torch.where order, measurements.py:217-219).'''

    def __init__(self, device, sigma_s, mask_opt):
        size = mask_opt['image_size']
        self.in_shape = (1, 3, size, size)
        self._make_ctx(L.OP_INPAINT, size, 1, sigma_s, device)
        self.sigma_s = torch.Tensor([sigma_s]).to(device)
        mask_cpu = self.generate_mask(mask_opt)
        self._install_mask(mask_cpu)
        self.pre_calculated = None

    def _install_mask(self, mask_cpu):
        m = np.ascontiguousarray(mask_cpu.numpy().astype(np.float32).reshape(3, *self.in_shape[-2:]))
        L.check(self.lib.kdip_op_set_mask(self._h, C.c_void_p(m.ctypes.data)))
        self.mask = mask_cpu.to(self.device)
        self._idx = torch.nonzero(mask_cpu.reshape(-1) > 0).reshape(-1).to(self.device)     # int64, ascending = (c,h,w) order

    def forward(self, data: torch.Tensor, flatten=False, noiseless=False):
        y = self._check(data)
        if not noiseless:
            y = y + self.sigma_s * _randn_like(y)                 # noise BEFORE masking (measurements.py:212-215)
        y = self._apply(y, False, y.shape)
        if flatten:
            B = y.shape[0]
            flat = torch.empty(B, self._idx.numel(), device=y.device)
            L.check(self.lib.kdip_gather(L.stream(), L.ptr(y), L.ptr(self._idx), self._idx.numel(), y[0].numel(), B, L.ptr(flat)))
            return y, flat
        return y

    def transpose(self, data, flatten=False):
        y = self._check(data)
        if flatten:
            B = y.shape[0]
            per = int(np.prod(self.in_shape[-3:]))
            x = torch.empty(B, *self.in_shape[-3:], device=y.device)
            L.check(self.lib.kdip_scatter(L.stream(), L.ptr(y), L.ptr(self._idx), self._idx.numel(), per, B, L.ptr(x)))
            return x
        return y.clone()

    def forward_adjoint(self, r):
        return self._apply(r, False, r.shape)

    def generate_mask(self, mask_opt):
        # the reference draws torch.randn(*in_shape) here (measurements.py:242): only its shape is used, but the draw advances
        # the torch CPU generator, so it is kept to leave the random stream of everything that follows unchanged
        return MaskGenerator(**mask_opt)(torch.randn(*self.in_shape))


class MaskGenerator:
    def __init__(self, mask_type, mask_len_range=None, mask_prob_range=None, image_size=256, margin=(16, 16)):
        assert mask_type in ['box', 'random', 'both', 'extreme']
        self.mask_type = mask_type
        self.mask_len_range = mask_len_range
        self.mask_prob_range = mask_prob_range
        self.image_size = image_size
        self.margin = margin

    def __call__(self, img):
        if self.mask_type == 'random':
            return self._retrieve_random(img)
        if self.mask_type in ('box', 'extreme'):
            mask = self._retrieve_box(img)
            return 1. - mask if self.mask_type == 'extreme' else mask
        raise NotImplementedError("mask_type 'both' has no branch in the reference's MaskGenerator.__call__ either (it returns None)")

    def _retrieve_box(self, img):
        """measurements.py:275-284,300-320: box height then width from np.random.randint(l, h) (global stream), box centred
        between the margins (the reference's random placement is commented out)."""
        l, h = int(self.mask_len_range[0]), int(self.mask_len_range[1])
        mask_h = np.random.randint(l, h)
        mask_w = np.random.randint(l, h)
        S = self.image_size
        t = (self.margin[0] + (S - self.margin[0] - mask_h)) // 2
        lft = (self.margin[1] + (S - self.margin[1] - mask_w)) // 2
        mask = torch.ones(img.shape[0], 3, S, S)
        mask[..., t:t + mask_h, lft:lft + mask_w] = 0
        return mask

    def _retrieve_random(self, img):
        """Draw order pinned to the reference (measurements.py:286-298): one np.random.uniform,
        then np.random.choice(S*S, int(S*S*p), replace=False) from the global MT19937 stream."""
        total = self.image_size ** 2
        l, h = self.mask_prob_range
        prob = np.random.uniform(l, h)
        keep = np.ones(total, dtype=np.float32)
        drop = np.random.choice(total, int(total * prob), replace=False)
        keep[drop] = 0
        plane = torch.from_numpy(keep).view(1, self.image_size, self.image_size)
        return plane.repeat(3, 1, 1)[None].expand(img.shape[0], -1, -1, -1).clone()
