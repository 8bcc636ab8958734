"""Debug: is the first guided call of a process different from the second (same inputs, fresh handles)?"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import smooth_image, synthetic_recon_mse
import kdip_amd.unet as ku, kdip_amd.measurements as km, kdip_amd.condition as kc
sigma_v = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
sd = ku.synthetic_state_dict(seed=0, **ku.IMAGENET_CONFIG)
opkw = dict(in_shape=(1, 3, 256, 256), kernel_size=61, intensity=0.5, sigma_s=0.05)
x0 = smooth_image(2, 256, 1)
x = x0 + sigma_v * torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(11))
rm = synthetic_recon_mse(); rmd = {k: v.cuda() for k, v in rm.items()}
outs = []
for rep in range(4):
    np.random.seed(0)
    hop = km.get_operator("motion_blur", device="cuda", **opkw)
    torch.manual_seed(2)
    meas = hop.forward(x0.clone().cuda(), flatten=True)
    m = ku.UNetModel(dtype="f32", **ku.IMAGENET_CONFIG); m.load_state_dict(sd)
    D = ku.GaussianDiffusionTables()
    hm = kc.ConditionOpenAIDenoiser(inner_model=m, diffusion=D, x0_cov_type="analytic", recon_mse=rmd, operator=hop, measurement=meas, guidance="I", device="cuda")
    sig = torch.full((2,), sigma_v, device="cuda")
    raw, _, _ = m.forward_raw(x.cuda(), torch.full((2,), 300.0, device="cuda"))
    inter = []
    def traced(xx, ss):
        import ctypes as C
        import kdip_amd._lib as L
        def cks():
            v = C.c_ulonglong(0); L.check(L.load().kdip_unet_debug_stash_checksum(m._h, L.stream(), C.byref(v))); return v.value
        x0_mean, x0_var, th = hm.uncond_pred(xx, ss)
        c0 = cks()
        raw_ = hm._stash[0][0].cpu(); dd = (raw_.abs() - 1).abs(); k = int(dd.argmin()); print(f"    image 0: x0_raw closest to the clamp boundary: | |x0_raw| - 1 | = {float(dd.min()):.3e} at {np.unravel_index(k, dd.shape)}, x0_raw = {float(raw_.flatten()[k]):.8f}")
        mat = hm._solve(x0_mean, x0_var, th)
        c1 = cks()
        ls = hm._vjp_x0(mat)
        c2 = cks()
        print(f"    stash checksum after forward {c0:x}, after solve {c1:x}, after vjp {c2:x}", "CHANGED" if len({c0, c1, c2}) > 1 else "")
        ls2 = hm._vjp_x0(mat)
        inter.append((x0_mean.cpu(), mat.cpu(), ls.cpu(), ls2.cpu()))
        return hm._combine(x0_mean, ls, float(ss[0]) ** 2)
    hat = traced(x.cuda(), sig).cpu()
    hat_b = traced(x.cuda(), sig).cpu()
    for nm, k in (("x0_mean", 0), ("mat", 1), ("vjp", 2)):
        print(f"  rep {rep}: {nm}: call1 vs call2 max|d| {float((inter[0][k] - inter[1][k]).abs().max()):.3e}")
    print(f"  rep {rep}: vjp repeated on the same saved state: call1 {float((inter[0][2] - inter[0][3]).abs().max()):.3e}  call2 {float((inter[1][2] - inter[1][3]).abs().max()):.3e}")
    outs.append((raw.cpu(), hat, hat_b, meas[0].cpu() if isinstance(meas, tuple) else meas.cpu()))
    del m, hm; torch.cuda.empty_cache()
for i in range(1, 4):
    print(f"rep {i} vs rep 0: raw max|d| {float((outs[i][0] - outs[0][0]).abs().max()):.3e}  hat {float((outs[i][1] - outs[0][1]).abs().max()):.3e}  meas {float((outs[i][3] - outs[0][3]).abs().max()):.3e}")
for i in range(4):
    d = (outs[i][1] - outs[i][2]).abs()
    print(f"rep {i}: first vs second call of the same handle: max|d| {float(d.max()):.3e}, n > 1e-3: {int((d > 1e-3).sum())}")
d = (outs[0][1] - outs[1][1]).abs()
print("differing pixels (rep0 vs rep1) > 1e-3:", int((d > 1e-3).sum()), "of", d.numel(), "max at", np.unravel_index(int(d.argmax()), d.shape))

d = (outs[0][1] - outs[1][1]).abs()
for b in range(d.shape[0]):
    m = (d[b] > 1e-3).any(0)
    ys, xs = np.nonzero(m.numpy())
    if len(ys): print(f"image {b}: {len(ys)} differing pixels, rows {ys.min()}..{ys.max()}, cols {xs.min()}..{xs.max()}; max {float(d[b].max()):.3f}")
    else: print(f"image {b}: identical")
    if len(ys):
        hist = np.zeros((8, 8), int)
        for y, x_ in zip(ys, xs): hist[y // 32, x_ // 32] += 1
        print(hist)
