"""dtype "f16x3" sanity on the GPU box: (1) tiny UNet forward / VJP against f32 and bf16x3, (2) a forward whose input leaves the fp16 window is redone
bf16-headed and equals the bf16x3 handle's result bit for bit, (3) one full-size FFHQ guided call (Type-I + Convert) f16x3 vs bf16x3 vs f32."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import kdip_amd.unet as ku, kdip_amd.sampling as ks
from oracle import unet as ounet
cfg = ounet.UNetConfig(**ounet.TINY); sd = ounet.init_state_dict(cfg, seed=0)
kw = dict(image_size=64, model_channels=32, num_res_blocks=1, attention_resolutions="32", channel_mult=(1, 2))
ms = {d: ku.UNetModel(dtype=d, **kw).load_state_dict(sd) for d in ("f32", "bf16x3", "f16x3")}
g = torch.Generator().manual_seed(4)
x = torch.randn(2, 3, 64, 64, generator=g).cuda(); t = torch.tensor([100.0, 700.0]).cuda(); cot = torch.randn(2, 6, 64, 64, generator=g).cuda()
o = {d: m.forward_raw(x, t)[0].clone() for d, m in ms.items()}; v = {d: m.vjp(cot).clone() for d, m in ms.items()}
for d in ("bf16x3", "f16x3"):
    print(d, "fwd rel-max vs f32 %.2e  vjp rel-max %.2e" % (float((o[d] - o["f32"]).abs().max() / o["f32"].abs().max()), float((v[d] - v["f32"]).abs().max() / v["f32"].abs().max())))
print("fallbacks so far", ms["f16x3"].x3_fallbacks)
ob = ms["bf16x3"].forward_raw(x * 1e7, t)[0].clone(); ms["bf16x3"].x3_saturated()
oh = ms["f16x3"].forward_raw(x * 1e7, t)[0].clone()
print("out-of-window forward: fallbacks", ms["f16x3"].x3_fallbacks, "degraded", ms["f16x3"].x3_degraded, "bitwise equal to bf16x3:", torch.equal(ob, oh), "finite", bool(torch.isfinite(oh).all()))
vb = ms["bf16x3"].vjp(cot).clone(); vh = ms["f16x3"].vjp(cot).clone()
print("vjp after it: equal", torch.equal(vb, vh), "rel diff %.2e" % float((vb - vh).abs().max() / vb.abs().max()), "fallbacks", ms["f16x3"].x3_fallbacks)
o2 = ms["f16x3"].forward_raw(x, t)[0]
print("back in window: equal to first f16x3 forward", torch.equal(o2, o["f16x3"]), "fallbacks", ms["f16x3"].x3_fallbacks)
del ms
WL = bench.WORKLOADS["cfg1"]; D = ku.GaussianDiffusionTables(); sdF = ku.synthetic_state_dict(seed=0, **ku.FFHQ_CONFIG)
sig = ks.get_sigmas_karras(100, 0.01, 80, rho=7.0, device="cpu"); outs = {}
for d in ("f32", "bf16x3", "f16x3"):
    den, op, x0, meas = bench.build_problem(WL, d, torch.device("cuda", 0), 2, sdF, D, seed=0)
    noise = torch.randn(2, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    for i in (10, 95):
        s = float(sig[i]); xx = (x0 + s * noise).contiguous()
        outs[(d, i)] = den(xx, torch.full((2,), s, device="cuda")).clone()
    if d == "f16x3": print("FFHQ f16x3 fallbacks", bench.unet_of(den).x3_fallbacks)
    del den
    torch.cuda.empty_cache()
for i in (10, 95):
    print("FFHQ guided call step", i, " max-abs vs f32: bf16x3 %.2e  f16x3 %.2e" % (float((outs[("bf16x3", i)] - outs[("f32", i)]).abs().max()), float((outs[("f16x3", i)] - outs[("f32", i)]).abs().max())))
