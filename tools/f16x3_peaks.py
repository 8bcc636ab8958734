"""Per-launch operand peaks of fp16-headed (dtype "f16x3") passes on the bench workloads: how far the conv operands of real guided calls sit from
the two sides of the fp16 window (65504 above, 2^-14 = 6.1e-5 below; the low-side flag fires under 2^-12 = 2.4e-4).
usage: python tools/f16x3_peaks.py [cfg1 cfg3 ...]"""
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import kdip_amd._lib as L, kdip_amd.unet as ku, kdip_amd.sampling as ks
lib = L.load()
def peaks(u):
    buf = (C.c_float * 1024)(); n = C.c_int(0)
    L.check(lib.kdip_debug_x3_peaks(u._h, L.stream(), buf, 1024, C.byref(n)))
    return np.array(buf[:n.value])
for wl in (sys.argv[1:] or ["cfg1"]):
    WL = bench.WORKLOADS[wl]; arch = ku.FFHQ_CONFIG if WL["arch"] == "FFHQ" else ku.IMAGENET_CONFIG
    sd = ku.synthetic_state_dict(seed=0, out_cov=bool(WL.get("ortho")), **arch); D = ku.GaussianDiffusionTables()
    B = min(WL["batch"], 4)
    den, op, x0, meas = bench.build_problem(WL, "f16x3", torch.device("cuda", 0), B, sd, D, seed=0)
    u = bench.unet_of(den); u.x3_guard = False
    sig = ks.get_sigmas_karras(WL["nsteps"], 0.01, 80, rho=7.0, device="cpu")
    noise = torch.randn(B, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    for i in sorted({0, WL["nsteps"] // 10, WL["nsteps"] // 2, WL["nsteps"] * 95 // 100, WL["nsteps"] - 1}):
        s = float(sig[i]); x = (x0 + s * noise).contiguous() if i else (noise * s).contiguous()
        den(x, torch.full((B,), s, device="cuda"))
        p = peaks(u); fl = u.x3_saturated()          # (last pass of the call: the VJP where the guidance has one)
        nz = p[p > 0]
        print(f"{wl} step {i} sigma {s:.3g}: last pass {len(p)} launches, peak min {nz.min():.3g} / median {np.median(nz):.3g} / max {nz.max():.3g}; below 2^-12: {int((nz < 2.0 ** -12).sum())}; flags {fl}; lowest five {np.sort(nz)[:5]}")
    del den
    torch.cuda.empty_cache()
