"""Per-block phase times of one conv3 launch (needs a -DC3_TIMING=1 build: build.py --variant timing -DC3_TIMING=1, run with
KDIP_LIB_PATH=.../libkdip_hip_timing.so).  usage: python tools/conv3_phases.py B Cin Cout H W [tf] [st] [res]"""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import kdip_amd._lib as L
from test_conv3_gpu import run_conv3
a = [int(v) for v in sys.argv[1:]]
B, Cin, Cout, H, W = a[:5]
tf = a[5] if len(a) > 5 else 0; stm = a[6] if len(a) > 6 else 0; res = a[7] if len(a) > 7 else 0
lib = L.load()
grid = ((B * (H // 8) * (W // 32) * (Cout // 128) + 7) // 8) * 8
buf = torch.zeros(grid, 8, dtype=torch.int64, device="cuda")
g = torch.Generator().manual_seed(0)
x = torch.randn(B, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5; b = torch.randn(Cout, generator=g)
kw = {}
if tf == 1: kw = dict(tf=1, tf_coef=torch.rand(B, Cin, 2, generator=g) + 0.5)
if tf == 2: kw = dict(tf=2, tf_coef=torch.rand(B, Cin, 4, generator=g) * 0.5 + 0.25, x2=torch.randn(B, Cin, H, W, generator=g))
if stm == 2: kw.update(stx=torch.randn(B, Cout, H, W, generator=g), st_coef=torch.rand(B, Cout, 2, generator=g) + 0.5, st_mr=torch.rand(B, 32, 2, generator=g) + 0.5)
run_conv3(x, w, b, Cout, res=torch.randn(B, Cout, H, W, generator=g) if res else None, st_mode=stm, reps=1, **kw)      # warm
L.check(lib.kdip_debug_conv3_timing(L.ptr(buf)))
run_conv3(x, w, b, Cout, res=torch.randn(B, Cout, H, W, generator=g) if res else None, st_mode=stm, reps=1, **kw)
L.check(lib.kdip_debug_conv3_timing(None))
t = buf.cpu().numpy().astype(np.int64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
us = lambda v: v / 100.0
pro, kl, epi, life = us(t[:, 1] - t[:, 0]), us(t[:, 2] - t[:, 1]), us(t[:, 3] - t[:, 2]), us(t[:, 3] - t[:, 0])
print(f"blocks {len(t)}; launch span {us(t[:, 3].max() - t0):.1f} us")
for name, v in (("prologue", pro), ("K loop", kl), ("epilogue", epi), ("block life", life)):
    print(f"  {name:10s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f} us")
# co-residency: blocks per CU key (xcc, se/sh/cu bits of HW_ID) and the phase offset between co-resident blocks
key = (t[:, 4] & 0xf) * 65536 + ((t[:, 5] >> 8) & 0xffff)
start = us(t[:, 0] - t0)
order = np.argsort(start)
print("  start-time histogram (us):", np.histogram(start, bins=12)[0].tolist(), "edges", [round(e) for e in np.histogram(start, bins=12)[1].tolist()])
cus = {}
for i in order: cus.setdefault(key[i], []).append((start[i], us(t[i, 3] - t0)))
print("  distinct CU keys:", len(cus))
k0 = list(cus.keys())[0]
print("  timeline of one CU (start, end):", [(round(a, 1), round(b, 1)) for a, b in cus[k0]][:12])
