"""Oracle: ADM UNet (guided_diffusion.unet.UNetModel) as a functional torch-CPU
fp32 forward over a reference-layout `state_dict`.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Autograd through this function is
the oracle for the hand-written HIP input-VJP.

Restricted to what the sampling scripts construct
(condition/diffpir_utils/utils_model.py:353-387 + configs/test_*.json):
`use_scale_shift_norm=True`, `resblock_updown=True`, `num_head_channels=64`
(legacy attention order), `learn_sigma=True`, no class conditioning.
"""
from dataclasses import dataclass, field
from typing import Tuple
import math
import torch
import torch.nn.functional as F

from .tables import timestep_embedding


@dataclass
class UNetConfig:
    image_size: int = 256
    in_channels: int = 3
    model_channels: int = 128
    out_channels: int = 6
    num_res_blocks: int = 1
    attention_resolutions: str = "16"
    channel_mult: Tuple[int, ...] = ()
    num_head_channels: int = 64

    def __post_init__(self):
        if not self.channel_mult:
            # guided_diffusion/script_util.py:148-160
            self.channel_mult = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4),
                                 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}[self.image_size]
        self.channel_mult = tuple(self.channel_mult)

    @property
    def attention_ds(self):
        # guided_diffusion/script_util.py:162-164
        return tuple(self.image_size // int(r) for r in self.attention_resolutions.split(","))


FFHQ = dict(image_size=256, model_channels=128, num_res_blocks=1, attention_resolutions="16")
IMAGENET = dict(image_size=256, model_channels=256, num_res_blocks=2, attention_resolutions="8,16,32")
TINY = dict(image_size=64, model_channels=32, num_res_blocks=1, attention_resolutions="32", channel_mult=(1, 2))


def unet_spec(cfg: UNetConfig):
    """Block list mirroring UNetModel.__init__ (guided_diffusion/unet.py:482-619).
    Each block = list of layers; layer = ('conv', cin, cout) | ('res', cin, cout, mode)
    | ('attn', ch), mode in {'none','down','up'}."""
    mc = cfg.model_channels
    ch = int(cfg.channel_mult[0] * mc)
    inp = [[("conv", cfg.in_channels, ch)]]
    chans = [ch]
    ds = 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            layers = [("res", ch, int(mult * mc), "none")]
            ch = int(mult * mc)
            if ds in cfg.attention_ds:
                layers.append(("attn", ch))
            inp.append(layers)
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            inp.append([("res", ch, ch, "down")])
            chans.append(ch)
            ds *= 2
    mid = [("res", ch, ch, "none"), ("attn", ch), ("res", ch, ch, "none")]
    out = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, int(mc * mult), "none")]
            ch = int(mc * mult)
            if ds in cfg.attention_ds:
                layers.append(("attn", ch))
            if level and i == cfg.num_res_blocks:
                layers.append(("res", ch, ch, "up"))
                ds //= 2
            out.append(layers)
    return inp, mid, out, ch


def param_shapes(cfg: UNetConfig, out_cov=False):
    """Ordered {reference state_dict key: shape}.  `zero` marks parameters that
    `zero_module` zeroes at construction (unet.py:210-212,295,617)."""
    mc = cfg.model_channels
    ted = mc * 4
    shapes = {}
    zero = set()

    def add(k, s, z=False):
        shapes[k] = tuple(s)
        if z:
            zero.add(k)

    add("time_embed.0.weight", (ted, mc)); add("time_embed.0.bias", (ted,))
    add("time_embed.2.weight", (ted, ted)); add("time_embed.2.bias", (ted,))

    def add_layers(prefix, layers):
        for j, L in enumerate(layers):
            p = f"{prefix}.{j}"
            if L[0] == "conv":
                add(f"{p}.weight", (L[2], L[1], 3, 3)); add(f"{p}.bias", (L[2],))
            elif L[0] == "res":
                _, cin, cout, _m = L
                add(f"{p}.in_layers.0.weight", (cin,)); add(f"{p}.in_layers.0.bias", (cin,))
                add(f"{p}.in_layers.2.weight", (cout, cin, 3, 3)); add(f"{p}.in_layers.2.bias", (cout,))
                add(f"{p}.emb_layers.1.weight", (2 * cout, ted)); add(f"{p}.emb_layers.1.bias", (2 * cout,))
                add(f"{p}.out_layers.0.weight", (cout,)); add(f"{p}.out_layers.0.bias", (cout,))
                add(f"{p}.out_layers.3.weight", (cout, cout, 3, 3), True); add(f"{p}.out_layers.3.bias", (cout,), True)
                if cin != cout:
                    add(f"{p}.skip_connection.weight", (cout, cin, 1, 1)); add(f"{p}.skip_connection.bias", (cout,))
            elif L[0] == "attn":
                c = L[1]
                add(f"{p}.norm.weight", (c,)); add(f"{p}.norm.bias", (c,))
                add(f"{p}.qkv.weight", (3 * c, c, 1)); add(f"{p}.qkv.bias", (3 * c,))
                add(f"{p}.proj_out.weight", (c, c, 1), True); add(f"{p}.proj_out.bias", (c,), True)

    inp, mid, out, ch = unet_spec(cfg)
    for i, layers in enumerate(inp):
        add_layers(f"input_blocks.{i}", layers)
    add_layers("middle_block", mid)
    for i, layers in enumerate(out):
        add_layers(f"output_blocks.{i}", layers)
    add("out.0.weight", (ch,)); add("out.0.bias", (ch,))
    add("out.2.weight", (cfg.out_channels, ch, 3, 3), True); add("out.2.bias", (cfg.out_channels,), True)
    if out_cov:
        # OpenAIDenoiserV2.out_cov = Conv2d(128, 6, 1) (k_diffusion/external.py:141)
        add("out_cov.weight", (6, ch, 1, 1)); add("out_cov.bias", (6,))
    return shapes, zero


def init_state_dict(cfg: UNetConfig, seed=0, out_cov=False, head_scale=1.0):
    """Seeded synthetic weights (no checkpoint is obtainable).
    PyTorch-default-like uniform(+-1/sqrt(fan_in)); zero-initialised modules are
    re-drawn N(0, 0.02^2) so outputs are not identically zero;
    GroupNorm affine perturbed so gamma/beta paths are exercised."""
    g = torch.Generator().manual_seed(seed)
    shapes, zero = param_shapes(cfg, out_cov)
    sd = {}
    for k, s in shapes.items():
        is_norm = (".in_layers.0." in k or ".out_layers.0." in k or ".norm." in k or k.startswith("out.0."))
        if is_norm:
            r = torch.randn(s, generator=g) * 0.1
            sd[k] = (1.0 + r) if k.endswith("weight") else r
        elif k in zero:
            sd[k] = torch.randn(s, generator=g) * 0.02
        else:
            if k.endswith("weight"):
                fan_in = 1
                for d in s[1:]:
                    fan_in *= d
            else:
                wk = k[:-4] + "weight"
                fan_in = 1
                for d in shapes[wk][1:]:
                    fan_in *= d
            b = 1.0 / math.sqrt(fan_in)
            sd[k] = (torch.rand(s, generator=g) * 2 - 1) * b
    if head_scale != 1.0:
        sd["out.2.weight"] = sd["out.2.weight"] * head_scale
        sd["out.2.bias"] = sd["out.2.bias"] * head_scale
    return sd


def _gn(x, w, b):
    # GroupNorm32(32, C) computed in fp32 (guided_diffusion/nn.py:17-19,93-100)
    return F.group_norm(x.float(), 32, w, b, eps=1e-5).type(x.dtype)


def _resblock(sd, p, x, emb, mode):
    """ResBlock._forward with use_scale_shift_norm (unet.py:237-257)."""
    h = F.silu(_gn(x, sd[f"{p}.in_layers.0.weight"], sd[f"{p}.in_layers.0.bias"]))
    if mode == "down":
        h = F.avg_pool2d(h, 2, 2); x = F.avg_pool2d(x, 2, 2)          # Downsample(use_conv=False) :133-137
    elif mode == "up":
        h = F.interpolate(h, scale_factor=2, mode="nearest")            # Upsample(use_conv=False) :101-108
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    h = F.conv2d(h, sd[f"{p}.in_layers.2.weight"], sd[f"{p}.in_layers.2.bias"], padding=1)
    emb_out = F.linear(F.silu(emb), sd[f"{p}.emb_layers.1.weight"], sd[f"{p}.emb_layers.1.bias"])
    scale, shift = torch.chunk(emb_out[..., None, None], 2, dim=1)
    h = _gn(h, sd[f"{p}.out_layers.0.weight"], sd[f"{p}.out_layers.0.bias"]) * (1 + scale) + shift
    h = F.conv2d(F.silu(h), sd[f"{p}.out_layers.3.weight"], sd[f"{p}.out_layers.3.bias"], padding=1)
    k = f"{p}.skip_connection.weight"
    if k in sd:
        x = F.conv2d(x, sd[k], sd[f"{p}.skip_connection.bias"])
    return x + h


def _attention(sd, p, x, head_ch):
    """AttentionBlock._forward + QKVAttentionLegacy (unet.py:301-307,339-356)."""
    b, c, hh, ww = x.shape
    x = x.reshape(b, c, -1)
    qkv = F.conv1d(_gn(x, sd[f"{p}.norm.weight"], sd[f"{p}.norm.bias"]), sd[f"{p}.qkv.weight"], sd[f"{p}.qkv.bias"])
    nh = c // head_ch
    length = qkv.shape[-1]
    ch = qkv.shape[1] // (3 * nh)
    q, k, v = qkv.reshape(b * nh, ch * 3, length).split(ch, dim=1)
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1).type(w.dtype)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, length)
    h = F.conv1d(a, sd[f"{p}.proj_out.weight"], sd[f"{p}.proj_out.bias"])
    return (x + h).reshape(b, c, hh, ww)


def _run_layers(sd, prefix, layers, h, emb, cfg):
    for j, L in enumerate(layers):
        p = f"{prefix}.{j}"
        if L[0] == "conv":
            h = F.conv2d(h, sd[f"{p}.weight"], sd[f"{p}.bias"], padding=1)
        elif L[0] == "res":
            h = _resblock(sd, p, h, emb, L[3])
        else:
            h = _attention(sd, p, h, cfg.num_head_channels)
    return h


def unet_forward(sd, cfg: UNetConfig, x, timesteps, return_feature=False):
    """UNetModel.forward (guided_diffusion/unet.py:636-668)."""
    inp, mid, out, _ = unet_spec(cfg)
    emb = timestep_embedding(timesteps, cfg.model_channels)
    emb = F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    hs = []
    h = x
    for i, layers in enumerate(inp):
        h = _run_layers(sd, f"input_blocks.{i}", layers, h, emb, cfg)
        hs.append(h)
    h = _run_layers(sd, "middle_block", mid, h, emb, cfg)
    for i, layers in enumerate(out):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_layers(sd, f"output_blocks.{i}", layers, h, emb, cfg)
    o = F.silu(_gn(h, sd["out.0.weight"], sd["out.0.bias"]))
    o = F.conv2d(o, sd["out.2.weight"], sd["out.2.bias"], padding=1)
    if return_feature:
        return o, h
    return o
