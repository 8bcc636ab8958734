"""UNet handle + DDPM tables (host side).

Mirrors, for the hot path only:
  * guided_diffusion.unet.UNetModel.forward(x, timesteps, y=None, return_feature=False)
    (guided_diffusion/unet.py:636-668) -- backed by kdip_unet_forward / kdip_unet_vjp;
  * script_util.create_model_and_diffusion (guided_diffusion/script_util.py:74-184) with the
    defaults of condition/diffpir_utils/utils_model.py:353-387;
  * the float64 tables of GaussianDiffusion / SpacedDiffusion
    (guided_diffusion/gaussian_diffusion.py:118-169, respace.py:63-91).
"""
import ctypes as C
import math
import numpy as np
import torch

from . import _lib as L


class GaussianDiffusionTables:
    """float64 numpy tables; attribute names match GaussianDiffusion."""

    def __init__(self, steps=1000):
        scale = 1000 / steps
        base = np.linspace(scale * 0.0001, scale * 0.02, steps, dtype=np.float64)   # gaussian_diffusion.py:27-35
        base_ac = np.cumprod(1.0 - base, axis=0)
        last, nb = 1.0, []
        for ac in base_ac:                                # respace.py:71-80 (identity respacing still re-derives betas)
            nb.append(1 - ac / last)
            last = ac
        betas = np.array(nb, dtype=np.float64)
        self.num_timesteps = steps
        self.betas = betas
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.log_betas = np.log(betas)

    def f32(self, name, t):
        """_extract_into_tensor semantics: float64 table entry cast to fp32 (gaussian_diffusion.py:904)."""
        return float(np.float32(getattr(self, name)[int(t)]))


def _channel_mult_default(image_size):
    # guided_diffusion/script_util.py:148-160
    return {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}[image_size]


class UNetModel:
    """Device-resident ADM UNet.  Not an nn.Module: weights live in MFMA fragment order
    inside libkdip_hip; `load_state_dict` takes the reference checkpoint layout."""

    def __init__(self, image_size=256, model_channels=128, num_res_blocks=1, attention_resolutions="16",
                 channel_mult="", num_head_channels=64, in_channels=3, out_channels=6, dtype="bf16", device=None):
        L.require_gpu()
        self.lib = L.load()
        if channel_mult in ("", None, ()):
            cm = _channel_mult_default(image_size)
        elif isinstance(channel_mult, str):
            cm = tuple(int(c) for c in channel_mult.split(","))
        else:
            cm = tuple(channel_mult)
        if any(int(c) != c for c in cm):
            raise ValueError("fractional channel_mult is not supported")
        self.image_size, self.model_channels, self.out_channels, self.in_channels = image_size, model_channels, out_channels, in_channels
        self.channel_mult = tuple(int(c) for c in cm)
        self.attention_ds = tuple(image_size // int(r) for r in str(attention_resolutions).split(","))
        self.dtype = dtype
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())       # explicit index: handles / workspaces live on ONE device
        self.device = dev
        h = C.c_void_p()
        ads = (C.c_int * len(self.attention_ds))(*self.attention_ds)
        cms = (C.c_int * len(self.channel_mult))(*self.channel_mult)
        if dtype not in L.DTYPES:
            raise ValueError(f"dtype must be one of {sorted(L.DTYPES)} (got {dtype!r})")
        L.check(self.lib.kdip_unet_create(dev.index, L.DTYPES[dtype], image_size, in_channels,
                                          model_channels, out_channels, num_res_blocks, ads, len(self.attention_ds),
                                          cms, len(self.channel_mult), num_head_channels, C.byref(h)))
        self._h = h
        self._finalized = False
        self.has_out_cov = False
        # dtype "f16x3": the fp16-headed split arithmetic is exact only while every conv operand stays inside the fp16 window; the library
        # raises a flag otherwise and this wrapper redoes the flagged call in the bf16-headed arithmetic (the handle carries both weight
        # encodings).  x3_guard = False leaves the polling to the caller (x3_saturated(), e.g. under hipGraph capture: graphs.py).
        self.x3_guard = dtype == "f16x3"
        self.x3_fallbacks = 0          # calls redone bf16-headed
        self.x3_degraded = 0           # ... of which the bf16-headed pass itself lost tail precision on some operand (bit 0 of the flag)
        # A VJP whose gradient tensors sit under the window (whole launches below 2^-12 of the cotangent's scale: the ImageNet-256
        # architecture at high sigma) is a property of the network, not of one input.  x3_auto_window = True: after the first such call
        # every dgrad launch takes its own scale (set_x3_window("launch"), +2 - 3 % per call) instead of a bf16-headed redo call after call.
        # Off by default: the switch makes a result depend on the handle's history (two runs of one schedule are no longer bitwise equal);
        # call set_x3_window("launch") up front for such networks, or use dtype "bf16x3".
        self.x3_auto_window = False
        self._x3_window = "vjp"
        self._x3_window_pending = False
        self.x3_window_switches = 0

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self.lib.kdip_unet_destroy(h)
            self._h = None

    def load_state_dict(self, state_dict, strict=True):
        """Accepts the reference key layout (input_blocks.N.M..., middle_block..., output_blocks...,
        out..., time_embed...) plus optional out_cov.* (OpenAIDenoiserV2)."""
        for k, v in state_dict.items():
            a = v.detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_long * a.ndim)(*a.shape)
            L.check(self.lib.kdip_unet_load(self._h, k.encode(), C.c_void_p(a.data_ptr()), shape, a.ndim))
            if k.startswith("out_cov."):
                self.has_out_cov = True
        L.check(self.lib.kdip_unet_finalize(self._h))     # raises on missing / mis-shaped parameters
        self._finalized = True
        return self

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def forward_raw(self, x, t, in_scale=1.0, want_cov=False, want_feature=False):
        """x [B,3,S,S] fp32 cuda, t [B] float.  Returns (out[B,6,S,S], cov|None, feature|None)."""
        assert x.is_cuda and x.dtype == torch.float32
        B = x.shape[0]
        x = x.contiguous()
        t = t.to(device=x.device, dtype=torch.float32).contiguous()
        out = torch.empty(B, self.out_channels, self.image_size, self.image_size, device=x.device, dtype=torch.float32)
        cov = torch.empty(B, 6, self.image_size, self.image_size, device=x.device) if want_cov else None
        feat = torch.empty(B, self.channel_mult[0] * self.model_channels, self.image_size, self.image_size,
                           device=x.device) if want_feature else None
        self.before_forward()
        self.guarded(lambda: L.check(self.lib.kdip_unet_forward(self._h, L.stream(), L.ptr(x), L.ptr(t), B, float(in_scale), L.ptr(out),
                                                                L.ptr(cov), L.ptr(feat))), "forward")
        return out, cov, feat

    def before_forward(self):
        """(dtype "f16x3") apply a window-mode switch a flagged VJP asked for: the switch re-plans the workspace and drops the activation
        stash, so it waits for the next forward (a caller may run several VJPs on one forward)."""
        if self._x3_window_pending and not torch.cuda.is_current_stream_capturing():
            self._x3_window_pending = False
            self.set_x3_window("launch")
            self.x3_window_switches += 1

    def guard_active(self):
        return self.x3_guard and self.dtype == "f16x3" and not torch.cuda.is_current_stream_capturing()

    def guarded(self, call, kind="call"):
        """Run `call` (library launches on the current stream that write their results in place).  dtype "f16x3": poll the fp16-window
        flag afterwards (one stream synchronisation) and, if some operand left the window, run `call` again in the bf16-headed
        arithmetic (kdip_unet_x3_head) -- the result a "bf16x3" handle would have produced."""
        call()
        if not self.guard_active():
            return
        flag = self.x3_saturated()
        if flag & 5:                      # bit 0: an operand above the fp16 window, bit 2: a launch's whole operand tensor below it
            self.x3_fallbacks += 1
            if (flag & 4) and kind != "forward" and self.x3_auto_window and self._x3_window == "vjp":
                self._x3_window_pending = True
            L.check(min(self.lib.kdip_unet_x3_head(self._h, 1), 0))
            try:
                call()
                self.x3_degraded += self.x3_saturated() & 1      # (also clears the flag for the next fp16-headed call)
            finally:
                L.check(min(self.lib.kdip_unet_x3_head(self._h, 0), 0))

    def set_x3_window(self, mode):
        """dtype "bf16x3" / "f16x3" only: "vjp" (default) = one fp16-window scale per VJP from max |cotangent|; "launch" = every dgrad conv scales
        by a sampled max of its own input (robust to networks with very large backward gains, +2 % per call)."""
        if mode not in ("vjp", "launch"):
            raise ValueError("mode must be 'vjp' or 'launch'")
        L.check(self.lib.kdip_unet_x3_window(self._h, 1 if mode == "launch" else 0))
        self._x3_window = mode

    def x3_saturated(self, reset=True):
        """dtype "bf16x3" / "f16x3" only: flags of operands that left the fp16 window of the split-precision convs since the last reset (bit 0:
        an activation / gradient operand in some launch, bit 1: a weight at pack time); 0 = every conv product carried its full
        precision.  One stream synchronisation."""
        import ctypes as C
        f = C.c_int(0)
        L.check(self.lib.kdip_unet_x3_saturated(self._h, L.stream(), 1 if reset else 0, C.byref(f)))
        return int(f.value)

    def set_deterministic(self, on=True):
        """dtype "f32" / "bf16x3" only (default on): fixed-order cross-block reductions, two runs of one call are bitwise equal
        (csrc/det.h).  Off = the floating-point atomics of the bf16 mode (A/B timing).  Returns the previous setting."""
        rc = self.lib.kdip_unet_deterministic(self._h, 1 if on else 0)
        if rc < 0:
            L.check(rc)
        return bool(rc)

    def vjp(self, cot):
        """(d out / d x_in)^T cot for the last forward; cot [B,6,S,S] -> [B,3,S,S]."""
        if not (cot.is_cuda and cot.dtype == torch.float32 and cot.device == self.device):
            raise L.KdipError(f"vjp: cotangent must be a float32 tensor on {self.device} (got {cot.dtype} on {cot.device})")
        if tuple(cot.shape[1:]) != (self.out_channels, self.image_size, self.image_size):
            raise L.KdipError(f"vjp: cotangent shape {tuple(cot.shape)} does not match [B, {self.out_channels}, {self.image_size}, {self.image_size}]")
        cot = cot.contiguous()
        B = cot.shape[0]
        gx = torch.empty(B, self.in_channels, self.image_size, self.image_size, device=cot.device)
        self.guarded(lambda: L.check(self.lib.kdip_unet_vjp(self._h, L.stream(), L.ptr(cot), B, L.ptr(gx))), "vjp")   # the library rejects B != batch of the last forward
        return gx

    def forward(self, x, timesteps, y=None, return_feature=False):
        assert y is None, "class conditioning is not on the sampler path"
        out, _, feat = self.forward_raw(x, timesteps, want_feature=return_feature)
        return (out, feat) if return_feature else out

    __call__ = forward

    def workspace_bytes(self, B):
        return int(self.lib.kdip_unet_workspace_bytes(self._h, B))

    def workspace_generation(self):
        """changes whenever the handle re-allocates its workspace arenas (captured hipGraphs of earlier calls are stale then)"""
        return int(self.lib.kdip_unet_workspace_generation(self._h))


def normalize_state_dict(obj, prefer_ema=True):
    """Checkpoint payload -> flat UNet `state_dict` in the reference key layout.

    * plain OpenAI checkpoints (`diffusion_ffhq_10m.pt`, `256x256_diffusion_uncond.pt`): returned as is;
    * Lightning checkpoints of the DWT-Var / DCT-Var models (`ffhq_dwt.ckpt`, train_openai.py:86-87): the `state_dict`
      entry with keys `model_ema.inner_model.*` / `model_ema.out_cov.*` (EMA copy, what
      `OpenAIDenoiser.load_from_checkpoint(...).model_ema` serves, sample_condition_openai_v2.py:121) or `model.*`;
      buffers that are not UNet parameters (sigmas, log_sigmas, EMA counters) are dropped.
    """
    sd = obj.get("state_dict", obj) if isinstance(obj, dict) else obj
    roots = ("model_ema.", "model.") if prefer_ema else ("model.", "model_ema.")
    for root in roots:
        if any(k.startswith(root + "inner_model.") for k in sd):
            out = {}
            for k, v in sd.items():
                if k.startswith(root + "inner_model."):
                    out[k[len(root + "inner_model."):]] = v
                elif k.startswith(root + "out_cov."):
                    out["out_cov." + k[len(root + "out_cov."):]] = v
            return out
    return dict(sd)


FFHQ_CONFIG = dict(image_size=256, model_channels=128, num_res_blocks=1, attention_resolutions="16")
IMAGENET_CONFIG = dict(image_size=256, model_channels=256, num_res_blocks=2, attention_resolutions="8,16,32")


def create_model_and_diffusion(image_size=256, num_channels=128, num_res_blocks=1, attention_resolutions="16",
                               channel_mult="", num_head_channels=64, learn_sigma=True, dtype="bf16", device=None,
                               diffusion_steps=1000, **unused):
    """script_util.create_model_and_diffusion for the sampler's configuration
    (resblock_updown, use_scale_shift_norm, learn_sigma, linear schedule, no respacing)."""
    for k, want in (("resblock_updown", True), ("use_scale_shift_norm", True), ("class_cond", False)):
        if k in unused and unused[k] != want:
            raise ValueError(f"{k}={unused[k]} is not supported on the MI355X path")
    model = UNetModel(image_size=image_size, model_channels=num_channels, num_res_blocks=num_res_blocks,
                      attention_resolutions=attention_resolutions, channel_mult=channel_mult,
                      num_head_channels=num_head_channels, out_channels=6 if learn_sigma else 3, dtype=dtype, device=device)
    return model, GaussianDiffusionTables(diffusion_steps)


# ---------------------------------------------------------------- synthetic weights ----
def _plan(image_size, model_channels, num_res_blocks, attention_ds, channel_mult, in_channels=3):
    """Block list of UNetModel.__init__ (guided_diffusion/unet.py:482-619): per block a list of
    ('conv', cin, cout) | ('res', cin, cout) | ('attn', ch)."""
    mc = model_channels
    ch = int(channel_mult[0] * mc)
    inp, chans, ds = [[("conv", in_channels, ch)]], [ch], 1
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks):
            layers = [("res", ch, int(mult * mc))]
            ch = int(mult * mc)
            if ds in attention_ds:
                layers.append(("attn", ch))
            inp.append(layers); chans.append(ch)
        if level != len(channel_mult) - 1:
            inp.append([("res", ch, ch)]); chans.append(ch); ds *= 2
    mid = [("res", ch, ch), ("attn", ch), ("res", ch, ch)]
    out = []
    for level, mult in list(enumerate(channel_mult))[::-1]:
        for i in range(num_res_blocks + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, int(mc * mult))]
            ch = int(mc * mult)
            if ds in attention_ds:
                layers.append(("attn", ch))
            if level and i == num_res_blocks:
                layers.append(("res", ch, ch)); ds //= 2
            out.append(layers)
    return inp, mid, out, ch


def state_dict_shapes(image_size=256, model_channels=128, num_res_blocks=1, attention_resolutions="16", channel_mult="",
                      out_channels=6, out_cov=False):
    """Ordered {reference state_dict key: shape} and the set of keys `zero_module` zeroes."""
    cm = _channel_mult_default(image_size) if channel_mult in ("", None, ()) else tuple(channel_mult)
    ads = tuple(image_size // int(r) for r in str(attention_resolutions).split(","))
    mc, ted = model_channels, model_channels * 4
    shapes, zero = {}, set()

    def add(k, s, z=False):
        shapes[k] = tuple(s)
        if z:
            zero.add(k)
    add("time_embed.0.weight", (ted, mc)); add("time_embed.0.bias", (ted,))
    add("time_embed.2.weight", (ted, ted)); add("time_embed.2.bias", (ted,))

    def add_layers(prefix, layers):
        for j, Lr in enumerate(layers):
            p = f"{prefix}.{j}"
            if Lr[0] == "conv":
                add(f"{p}.weight", (Lr[2], Lr[1], 3, 3)); add(f"{p}.bias", (Lr[2],))
            elif Lr[0] == "res":
                _, cin, cout = Lr
                add(f"{p}.in_layers.0.weight", (cin,)); add(f"{p}.in_layers.0.bias", (cin,))
                add(f"{p}.in_layers.2.weight", (cout, cin, 3, 3)); add(f"{p}.in_layers.2.bias", (cout,))
                add(f"{p}.emb_layers.1.weight", (2 * cout, ted)); add(f"{p}.emb_layers.1.bias", (2 * cout,))
                add(f"{p}.out_layers.0.weight", (cout,)); add(f"{p}.out_layers.0.bias", (cout,))
                add(f"{p}.out_layers.3.weight", (cout, cout, 3, 3), True); add(f"{p}.out_layers.3.bias", (cout,), True)
                if cin != cout:
                    add(f"{p}.skip_connection.weight", (cout, cin, 1, 1)); add(f"{p}.skip_connection.bias", (cout,))
            else:
                c = Lr[1]
                add(f"{p}.norm.weight", (c,)); add(f"{p}.norm.bias", (c,))
                add(f"{p}.qkv.weight", (3 * c, c, 1)); add(f"{p}.qkv.bias", (3 * c,))
                add(f"{p}.proj_out.weight", (c, c, 1), True); add(f"{p}.proj_out.bias", (c,), True)
    inp, mid, out, ch = _plan(image_size, mc, num_res_blocks, ads, cm)
    for i, layers in enumerate(inp):
        add_layers(f"input_blocks.{i}", layers)
    add_layers("middle_block", mid)
    for i, layers in enumerate(out):
        add_layers(f"output_blocks.{i}", layers)
    add("out.0.weight", (ch,)); add("out.0.bias", (ch,))
    add("out.2.weight", (out_channels, ch, 3, 3), True); add("out.2.bias", (out_channels,), True)
    if out_cov:
        add("out_cov.weight", (6, ch, 1, 1)); add("out_cov.bias", (6,))
    return shapes, zero


def synthetic_state_dict(seed=0, out_cov=False, **cfg):
    """Seeded random-init weights of the named architecture (no checkpoint is obtainable offline):
    uniform(+-1/sqrt(fan_in)) like PyTorch's default; zero-initialised modules re-drawn N(0, 0.02^2)
    so the network output is not identically zero; GroupNorm affine = 1 + 0.1 n, 0.1 n."""
    g = torch.Generator().manual_seed(seed)
    shapes, zero = state_dict_shapes(out_cov=out_cov, **cfg)
    sd = {}
    for k, s in shapes.items():
        if ".in_layers.0." in k or ".out_layers.0." in k or ".norm." in k or k.startswith("out.0."):
            r = torch.randn(s, generator=g) * 0.1
            sd[k] = (1.0 + r) if k.endswith("weight") else r
        elif k in zero:
            sd[k] = torch.randn(s, generator=g) * 0.02
        else:
            wshape = s if k.endswith("weight") else shapes[k[:-4] + "weight"]
            fan_in = int(np.prod(wshape[1:]))
            sd[k] = (torch.rand(s, generator=g) * 2 - 1) * (1.0 / math.sqrt(fan_in))
    return sd
