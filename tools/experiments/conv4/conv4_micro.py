"""A/B micro-benchmark conv3 vs conv4 through kdip_test_conv3 with the kernel generation forced: interleaved rounds in one process.
usage: python tools/conv4_micro.py B Cin Cout H W [tf 0|1|2] [st_mode 0|1|2] [res 0|1] [reps] [rounds]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from test_conv3_gpu import run_conv3
import kdip_amd._lib as L
a = [int(v) for v in sys.argv[1:]]
B, Cin, Cout, H, W = a[:5]
tf = a[5] if len(a) > 5 else 0
stm = a[6] if len(a) > 6 else 0
res = a[7] if len(a) > 7 else 0
reps = a[8] if len(a) > 8 else 30
rounds = a[9] if len(a) > 9 else 3
gens = [int(v) for v in os.environ.get("GENS", "3,4").split(",")]
g = torch.Generator().manual_seed(0)
x = torch.randn(B, Cin, H, W, generator=g)
w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
b = torch.randn(Cout, generator=g)
kw = {}
if tf == 1:
    kw = dict(tf=1, tf_coef=torch.rand(B, Cin, 2, generator=g) + 0.5)
elif tf == 2:
    kw = dict(tf=2, tf_coef=torch.rand(B, Cin, 4, generator=g) * 0.5 + 0.25, x2=torch.randn(B, Cin, H, W, generator=g))
if stm == 2:
    kw.update(stx=torch.randn(B, Cout, H, W, generator=g), st_coef=torch.rand(B, Cout, 2, generator=g) + 0.5, st_mr=torch.rand(B, 32, 2, generator=g) + 0.5)
r = torch.randn(B, Cout, H, W, generator=g) if res else None
fl = 2.0 * B * H * W * Cin * Cout * 9
lib = L.load()
out = {}
for rnd in range(rounds):
    for gen in gens:
        L.check(lib.kdip_debug_conv_generation(gen))
        y, sums, us = run_conv3(x, w, b, Cout, res=r, st_mode=stm, reps=reps, **kw)
        out.setdefault(gen, []).append(us)
        ys = out.setdefault(("y", gen), y)
L.check(lib.kdip_debug_conv_generation(0))
for gen in gens:
    us = sorted(out[gen])
    print(f"conv{gen} B={B} {Cin}->{Cout} @{H}x{W} tf={tf} st={stm} res={res}: min {us[0]:.1f} med {us[len(us) // 2]:.1f} us/launch, "
          f"{fl / us[len(us) // 2] / 1e6:.1f} TFLOP/s ({fl / us[len(us) // 2] / 1e6 / 2500:.3f} of 2.5 PF)")
if len(gens) == 2:
    d = (out[("y", gens[0])] - out[("y", gens[1])]).abs().max() / out[("y", gens[0])].abs().max()
    print(f"max rel diff between generations: {float(d):.2e}")
