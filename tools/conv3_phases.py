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
ntiles = B * (H // 8) * (W // 32) * (Cout // 128)
grid = min(((ntiles + 7) // 8) * 8, 512)          # persistent launch: 2 blocks per CU
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
pro, kl, epi, life = us(t[:, 1] - t[:, 0]), us(t[:, 2] - t[:, 1]), us(t[:, 3] - t[:, 2]), us(t[:, 6] - t[:, 0])
tpb = ntiles / len(t)
print(f"persistent blocks {len(t)} ({tpb:.2f} tiles each); launch span {us(t[:, 6].max() - t0):.1f} us; later tiles: {((life - us(t[:, 3] - t[:, 0])) / max(tpb - 1, 1e-9)).mean():.2f} us per tile")
st, en = us(t[:, 0] - t0), us(t[:, 6] - t0)
print(f"  block start after first block: p50 {np.percentile(st, 50):.2f} p90 {np.percentile(st, 90):.2f} max {st.max():.2f} us;  block end: p10 {np.percentile(en, 10):.2f} p50 {np.percentile(en, 50):.2f} p90 {np.percentile(en, 90):.2f} max {en.max():.2f} us")
e1, e2, e3 = us(t[:, 4] - t[:, 2]), us(t[:, 5] - t[:, 4]), us(t[:, 3] - t[:, 5])
for name, v in (("prologue 1", pro), ("K loop 1", kl), ("epilogue 1", epi), ("  sweep 1 (loads, pack, store)", e1), ("  sweep 2 (bwd stats)", e2), ("  stats combine", e3), ("    of which reduce-scatter", us(np.maximum(t[:, 7] - t[:, 5], 0))), ("block life", life)):
    print(f"  {name:10s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f} us")

if stm == 0:      # slots 7 / 5 = shader-clock counter (s_memtime) at block start / end: clock the CUs actually ran at
    cyc = (t[:, 5] - t[:, 7]).astype(np.float64)
    ghz = cyc / (life * 1e3)
    print(f"shader clock during the launch (s_memtime cycles / wall time per block): mean {ghz.mean():.3f} GHz, p10 {np.percentile(ghz, 10):.3f}, p90 {np.percentile(ghz, 90):.3f}")
