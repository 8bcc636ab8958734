"""Orthogonal basis transforms -- call surface of condition/utils.py:50-139.

OrthoTransform(None | 'dwt' | 'dct'): __call__(x) / .inv(x) on [B,3,S,S] fp32 device tensors.
'dwt' = Haar level-3 in PyWavelets' coeffs_to_array (Mallat) layout; 'dct' = orthonormal DCT-II
over (C,H,W) (the reference's dctn over all axes of a batch-1 tensor).  Both run on device
(the reference round-trips through numpy on the CPU per call).
"""
import ctypes as C
import torch

from . import _lib as L

__OT__ = dict()


def register_ot(name: str):
    def wrapper(cls):
        __OT__[name] = cls
        return cls
    return wrapper


_CODES = {None: L.OT_NONE, "dwt": L.OT_DWT, "dct": L.OT_DCT}
_ctx_cache = {}


def _ctx(size, code, device_index):
    """A bare operator context that only carries the basis (DCT matrix / scratch)."""
    key = (size, code, device_index)
    if key not in _ctx_cache:
        lib = L.load()
        h = C.c_void_p()
        L.check(lib.kdip_op_create(device_index, L.OP_INPAINT, size, 1, 0.0, C.byref(h)))
        L.check(lib.kdip_op_set_ortho(h, code))
        _ctx_cache[key] = h
    return _ctx_cache[key]


@register_ot('dct')
class DiscreteCosineTransform:
    code = L.OT_DCT


@register_ot('dwt')
class DiscreteWaveletTransform:
    code = L.OT_DWT


class OrthoTransform:
    def __init__(self, ortho_tf_type=None):
        if ortho_tf_type is not None and ortho_tf_type not in __OT__:
            raise KeyError(ortho_tf_type)
        self.ortho_tf_type = ortho_tf_type
        self.code = L.OT_NONE if ortho_tf_type is None else __OT__[ortho_tf_type].code

    def _run(self, x, inverse):
        if self.ortho_tf_type is None:
            return x
        L.require_gpu()
        assert x.is_cuda and x.dtype == torch.float32 and x.ndim == 4 and x.shape[1] == 3 and x.shape[2] == x.shape[3]
        x = x.contiguous()
        out = torch.empty_like(x)
        h = _ctx(x.shape[-1], self.code, x.device.index if x.device.index is not None else torch.cuda.current_device())
        L.check(L.load().kdip_op_ortho(h, L.stream(), L.ptr(x), x.shape[0], int(inverse), L.ptr(out)))
        return out

    def __call__(self, x: torch.Tensor):
        return self._run(x, False)

    def inv(self, x: torch.Tensor):
        return self._run(x, True)
