"""Micro-benchmark of conv3 (csrc/conv3.hip) through kdip_test_conv3: mean HIP-event time per launch and TFLOP/s for one shape
and fusion mode.  usage: python tools/conv3_micro.py B Cin Cout H W [tf 0|1|2] [st_mode 0|1|2] [res 0|1] [reps]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from test_conv3_gpu import run_conv3
a = [int(v) for v in sys.argv[1:]]
B, Cin, Cout, H, W = a[:5]
tf = a[5] if len(a) > 5 else 0
stm = a[6] if len(a) > 6 else 0
res = a[7] if len(a) > 7 else 0
reps = a[8] if len(a) > 8 else 20
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
y, sums, us = run_conv3(x, w, b, Cout, res=torch.randn(B, Cout, H, W, generator=g) if res else None, st_mode=stm, reps=reps, **kw)
fl = 2.0 * B * H * W * Cin * Cout * 9
print(f"conv3 B={B} {Cin}->{Cout} @{H}x{W} tf={tf} st={stm} res={res}: {us:.1f} us/launch, {fl / us / 1e6:.1f} TFLOP/s ({fl / us / 1e6 / 2500:.3f} of 2.5 PF)")
