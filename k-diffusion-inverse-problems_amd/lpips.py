"""LPIPS-VGG forward on the MI355X path -- `lpips.LPIPS(net='vgg')` as the reference's compute_metrics uses it
(sample_condition_openai.py:41-49,161; algorithm of the `lpips` package v0.1: lpips/lpips.py:LPIPS.forward,
lpips/pretrained_networks.py:vgg16): ScalingLayer, the 13 VGG-16 convs up to relu5_3 with features tapped after relu1_2 / 2_2 /
3_3 / 4_3 / 5_3, channel-unit-normalisation, squared difference, 1x1 `lin` layers, spatial mean, sum over layers.

Device work goes through libkdip_hip (kdip_conv_* = implicit-GEMM conv layers, kdip_relu_maxpool, kdip_lpips_layer).  The pretrained
weights (torchvision VGG-16 + the package's `vgg.pth` lin layers) are not obtainable offline: `load_state_dict` accepts the
`lpips.LPIPS` state_dict layout (`net.sliceK.N.weight`, `linK.model.1.weight`) or a torchvision `features.N.weight` layout plus lin
weights, and `synthetic_state_dict(seed)` draws seeded weights of the same shapes for the tests."""
import ctypes as C

import torch

from . import _lib as L

# (slice, index inside torchvision's vgg16().features, Cin, Cout); a 2x2 max pool precedes the first conv of slices 2..5
VGG_CONVS = [(1, 0, 3, 64), (1, 2, 64, 64), (2, 5, 64, 128), (2, 7, 128, 128), (3, 10, 128, 256), (3, 12, 256, 256), (3, 14, 256, 256),
             (4, 17, 256, 512), (4, 19, 512, 512), (4, 21, 512, 512), (5, 24, 512, 512), (5, 26, 512, 512), (5, 28, 512, 512)]
LIN_CHANNELS = [64, 128, 256, 512, 512]
SHIFT = (-.030, -.088, -.188)
SCALE = (.458, .448, .450)


def synthetic_state_dict(seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for sl, idx, cin, cout in VGG_CONVS:
        sd[f"net.slice{sl}.{idx}.weight"] = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (cin * 9)) ** 0.5
        sd[f"net.slice{sl}.{idx}.bias"] = torch.randn(cout, generator=g) * 0.05
    for k, c in enumerate(LIN_CHANNELS):
        sd[f"lin{k}.model.1.weight"] = torch.rand(1, c, 1, 1, generator=g) / c
    return sd


class LPIPS:
    """loss = LPIPS(...)(in0, in1): in0 / in1 [3,H,W] or [B,3,H,W] (H, W powers of two >= 128); returns [B,1,1,1] like the package."""

    def __init__(self, net="vgg", dtype="f32", device=None):
        if net != "vgg":
            raise ValueError("only net='vgg' is built (the reference's choice)")
        L.require_gpu()
        self.lib = L.load()
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.device, self.dtype = dev, dtype
        self._convs, self._lin = [], []

    def __del__(self):
        for h in getattr(self, "_convs", []):
            self.lib.kdip_conv_destroy(h)
        self._convs = []

    def load_state_dict(self, sd, lin=None):
        """`sd`: lpips.LPIPS layout, or torchvision vgg16 layout (`features.N.*`) with `lin` = {lin0.model.1.weight, ...}."""
        def conv_key(sl, idx, what):
            for k in (f"net.slice{sl}.{idx}.{what}", f"features.{idx}.{what}"):
                if k in sd:
                    return sd[k]
            raise KeyError(f"LPIPS: missing VGG parameter for features[{idx}].{what}")
        lin = lin if lin is not None else sd
        self.__del__()
        for sl, idx, cin, cout in VGG_CONVS:
            w = conv_key(sl, idx, "weight").detach().to("cpu", torch.float32).contiguous()
            b = conv_key(sl, idx, "bias").detach().to("cpu", torch.float32).contiguous()
            if tuple(w.shape) != (cout, cin, 3, 3):
                raise ValueError(f"LPIPS: features[{idx}].weight has shape {tuple(w.shape)}, expected {(cout, cin, 3, 3)}")
            h = C.c_void_p()
            L.check(self.lib.kdip_conv_create(self.device.index, L.DTYPES[self.dtype], C.c_void_p(w.data_ptr()),
                                              C.c_void_p(b.data_ptr()), cout, cin, 9, C.byref(h)))
            self._convs.append(h)
        self._lin = []
        for k, c in enumerate(LIN_CHANNELS):
            w = lin[f"lin{k}.model.1.weight"].detach().to(torch.float32).reshape(-1)
            if w.numel() != c:
                raise ValueError(f"LPIPS: lin{k} has {w.numel()} weights, expected {c}")
            self._lin.append(w.to(self.device).contiguous())
        return self

    def _features(self, x):
        B, _, H, W = x.shape
        feats, cur, h, w = [], x, H, W
        ws_bytes = max(self.lib.kdip_conv_workspace_bytes(hc, B, H, W) for hc in self._convs[:2])
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
        for (sl, idx, cin, cout), hc in zip(VGG_CONVS, self._convs):
            if idx in (5, 10, 17, 24):               # first conv of slices 2..5: 2x2 max pool of the (already rectified) features
                pooled = torch.empty(B, cin, h // 2, w // 2, device=self.device)
                L.check(self.lib.kdip_relu_maxpool(L.stream(), L.ptr(cur), B * cin, h, w, 1, L.ptr(pooled)))
                cur, h, w = pooled, h // 2, w // 2
            y = torch.empty(B, cout, h, w, device=self.device)
            L.check(self.lib.kdip_conv_apply(hc, L.stream(), L.ptr(cur), B, h, w, L.ptr(y), L.ptr(ws)))
            r = torch.empty_like(y)
            L.check(self.lib.kdip_relu_maxpool(L.stream(), L.ptr(y), B * cout, h, w, 0, L.ptr(r)))
            cur = r
            if idx in (2, 7, 14, 21, 28):            # relu1_2, relu2_2, relu3_3, relu4_3, relu5_3
                feats.append((cur, cout, h * w))
        return feats

    @staticmethod
    def check_size(H, W):
        """The VGG feature pyramid runs on the conv kernels' tiles (four 2x2 max-pools down to >= 8x8 maps): power-of-two sizes
        >= 128 only.  Callers that know their image size up front (sample_condition.py --lpips-checkpoint) call this BEFORE
        sampling, so that an unsupported size fails at start-up instead of after the samples were paid for."""
        if H < 128 or W < 128 or (H & (H - 1)) or (W & (W - 1)):
            raise L.KdipError(f"LPIPS: image size {H}x{W} is not supported (power-of-two sizes >= 128 only)")

    def __call__(self, in0, in1, normalize=False):
        if not self._convs:
            raise L.KdipError("LPIPS: load_state_dict first (the pretrained weights are not bundled)")
        if in0.dim() == 3:
            in0 = in0[None]
        if in1.dim() == 3:
            in1 = in1[None]
        try:                                          # lpips.LPIPS broadcasts its two inputs (e.g. one reference against a batch)
            in0, in1 = torch.broadcast_tensors(in0, in1)
        except RuntimeError:
            raise L.KdipError(f"LPIPS: input shapes {tuple(in0.shape)} and {tuple(in1.shape)} do not broadcast")
        self.check_size(*in0.shape[-2:])
        in0 = in0.to(self.device, torch.float32)
        in1 = in1.to(self.device, torch.float32)
        if normalize:                                 # [0,1] -> [-1,1] (the package's flag; the reference leaves it off)
            in0, in1 = 2 * in0 - 1, 2 * in1 - 1
        shift = torch.tensor(SHIFT, device=self.device).view(1, 3, 1, 1)
        scale = torch.tensor(SCALE, device=self.device).view(1, 3, 1, 1)
        B = in0.shape[0]
        both = torch.cat([(in0 - shift) / scale, (in1 - shift) / scale]).contiguous()      # ScalingLayer; one pass for both images
        feats = self._features(both)
        out = torch.zeros(B, device=self.device)
        for (f, c, hw), w in zip(feats, self._lin):
            L.check(self.lib.kdip_lpips_layer(L.stream(), L.ptr(f[:B].contiguous()), L.ptr(f[B:].contiguous()), L.ptr(w), B, c, hw, L.ptr(out)))
        return out.view(B, 1, 1, 1)
