"""Per-phase timeline of conv4 blocks (a -DC4_TIMING=1 build: tools/c4_variants.sh timing "-DC4_TIMING=1"; run with
KDIP_LIB_PATH=.../libkdip_hip_timing.so).  Stamps (shader cycles) of every wave of every block's first tile: per phase
t0 start of the load part, t1 in front of barrier X, t2 behind it (+ lgkmcnt(0)), t3 behind the MFMAs; prints medians over blocks.
usage: python tools/conv4_phases.py B Cin Cout H W [tf st res]"""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from test_conv3_gpu import run_conv3
import kdip_amd._lib as L
a = [int(v) for v in sys.argv[1:]]
B, Cin, Cout, H, W = a[:5]
tf, stm, res = (a[5:8] + [0, 0, 0])[:3] if len(a) > 5 else (0, 0, 0)
lib = L.load()
L.check(lib.kdip_debug_conv_generation(4))
g = torch.Generator().manual_seed(0)
x = torch.randn(B, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5; b = torch.randn(Cout, generator=g)
kw = {}
if tf == 1: kw = dict(tf=1, tf_coef=torch.rand(B, Cin, 2, generator=g) + 0.5)
elif tf == 2: kw = dict(tf=2, tf_coef=torch.rand(B, Cin, 4, generator=g) * 0.5 + 0.25, x2=torch.randn(B, Cin, H, W, generator=g))
if stm == 2: kw.update(stx=torch.randn(B, Cout, H, W, generator=g), st_coef=torch.rand(B, Cout, 2, generator=g) + 0.5, st_mr=torch.rand(B, 32, 2, generator=g) + 0.5)
r = torch.randn(B, Cout, H, W, generator=g) if res else None
run_conv3(x, w, b, Cout, res=r, st_mode=stm, **kw)            # warm-up without stamps
buf = torch.zeros(256 * 8 * 176, dtype=torch.int64, device="cuda")
L.check(lib.kdip_debug_conv3_timing(L.ptr(buf)))
run_conv3(x, w, b, Cout, res=r, st_mode=stm, **kw)
L.check(lib.kdip_debug_conv3_timing(None))
t = buf.cpu().numpy().reshape(256, 8, 176)
# tile-level stamps: [145 + 3 j + (0 K-loop start, 1 K-loop done, 2 epilogue done)], kernel start / end (cycles, 100 MHz) at 170..173
ok0 = t[:, 0, 170] > 0
tt = t[ok0].astype(np.float64)
clk = (tt[:, :, 172] - tt[:, :, 170]) / ((tt[:, :, 173] - tt[:, :, 171]) * 10.0)      # cycles per ns
print(f"blocks {int(ok0.sum())}; shader clock during the kernel: {np.median(clk):.3f} GHz; block life {np.median(tt[:, 0, 172] - tt[:, 0, 170]):.0f} cycles")
rt0, rt1 = tt[:, :, 171], tt[:, :, 173]
print(f"wall (100 MHz clock): first block start -> last block end {10 * (rt1.max() - rt0.min()) / 1e3:.1f} us; block start spread {10 * (rt0[:, 0].max() - rt0[:, 0].min()) / 1e3:.2f} us; "
      f"block life min / median / max {10 * np.min(rt1[:, 0] - rt0[:, 0]) / 1e3:.1f} / {10 * np.median(rt1[:, 0] - rt0[:, 0]) / 1e3:.1f} / {10 * np.max(rt1[:, 0] - rt0[:, 0]) / 1e3:.1f} us")
for wv in (0, 4):
    for j in range(8):
        a, b, c_ = tt[:, wv, 145 + 3 * j], tt[:, wv, 146 + 3 * j], tt[:, wv, 147 + 3 * j]
        if not (0 < np.median(b - a) < 1e8 and 0 < np.median(a - tt[:, wv, 170]) < 1e9): break
        nxt_ = tt[:, wv, 145 + 3 * (j + 1)] if j < 7 and 0 < np.median(tt[:, wv, 145 + 3 * (j + 1)] - c_) < 1e8 else None
        print(f"  wave {wv} tile {j}: start +{np.median(a - tt[:, wv, 170]):8.0f}  K loop {np.median(b - a):7.0f}  epilogue {np.median(c_ - b):7.0f}" + (f"  to next K loop {np.median(nxt_ - c_):6.0f}" if nxt_ is not None else ""))
for wv in (0, 4):
    e = tt[:, wv, 160:164]; kd = tt[:, wv, 146 + 3]; ed = tt[:, wv, 147 + 3]
    if np.median(e[:, 0]) > 0:
        print(f"  wave {wv} tile 1 epilogue: K loop done -> aux loads requested {np.median(e[:, 0] - kd):.0f}; sweep 1 (pack / fwd stats / stores) {np.median(e[:, 1] - e[:, 0]):.0f}; "
              f"sweep 2 math {np.median(e[:, 2] - e[:, 1]):.0f}; stores {np.median(e[:, 3] - e[:, 2]):.0f}; statistics reduce + atomics {np.median(ed - e[:, 3]):.0f}")
if os.environ.get("LEVEL", "2") == "1": sys.exit(0)
nph = min(36, (Cin // 32) * 9)
T = t[:, :, :nph * 4].reshape(256, 8, nph, 4).astype(np.float64)
ok = T[:, 0, 0, 0] > 0
T = T[ok]
print("blocks with stamps:", int(ok.sum()))
for grp, wv in (("group0 (wave 0)", 0), ("group1 (wave 4)", 4)):
    d_load = T[:, wv, :, 1] - T[:, wv, :, 0]
    d_barx = T[:, wv, :, 2] - T[:, wv, :, 1]
    d_mfma = T[:, wv, :, 3] - T[:, wv, :, 2]
    d_bary = np.concatenate([T[:, wv, 1:, 0] - T[:, wv, :-1, 3], np.full((T.shape[0], 1), np.nan)], axis=1)
    print(grp, "phase: load | barrier X | MFMA | barrier Y   (median cycles over blocks)")
    for ph in range(nph):
        print(f"  {ph:2d} (tap {ph % 9}): {np.median(d_load[:, ph]):7.0f} {np.median(d_barx[:, ph]):7.0f} {np.median(d_mfma[:, ph]):7.0f} {np.nanmedian(d_bary[:, ph]):7.0f}")
    per = (T[:, wv, nph - 1, 3] - T[:, wv, 0, 0]) / nph
    print(f"  mean cycles per phase: {np.median(per):.0f};  sums: load {np.median(d_load.sum(1)):.0f} barX {np.median(d_barx.sum(1)):.0f} mfma {np.median(d_mfma.sum(1)):.0f} barY {np.nanmedian(np.nansum(d_bary, 1)):.0f}")
# skew between waves of a group in front of barrier X
sk = T[:, 0:4, :, 1].max(1) - T[:, 0:4, :, 1].min(1)
print("arrival skew of group-0 waves in front of barrier X (median cycles):", float(np.median(sk)))
if os.environ.get("RAW"):
    for blk in (0, 100):
        base = T[blk, 0, 0, 0]
        for wv in range(8):
            print(f"block {blk} wave {wv}:")
            for ph in [int(v) for v in os.environ["RAW"].split(",")]:
                print("   ph %2d: " % ph + " ".join("%7.0f" % (T[blk, wv, ph, k] - base) for k in range(4)))
