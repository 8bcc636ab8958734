"""Operator-kernel bandwidths at 64 images (192 planes of 256 x 256 fp32), HIP events of the library profiler, per class and per tag.
usage: python tools/op_bw.py [images]"""
import ctypes as C, csv, collections, os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kdip_amd._lib as L, kdip_amd.measurements as km
from kdip_amd.transforms import OrthoTransform
from bench import smooth_image
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = L.load(); S = 256; dev = "cuda"
xb = smooth_image(n, S, seed=7).to(dev)
np.random.seed(0)
ops = {k: km.get_operator(k, device=dev, **kw) for k, kw in (
    ("gaussian_blur", dict(in_shape=(1, 3, S, S), kernel_size=61, intensity=3.0, sigma_s=0.05)),
    ("motion_blur", dict(in_shape=(1, 3, S, S), kernel_size=61, intensity=0.5, sigma_s=0.05)),
    ("super_resolution", dict(in_shape=(1, 3, S, S), scale_factor=4, sigma_s=0.05)),
    ("inpainting", dict(sigma_s=0.05, mask_opt=dict(mask_type="random", mask_prob_range=(0.5, 0.5), image_size=S))))}
dwt = OrthoTransform("dwt")
def run():
    yg = ops["gaussian_blur"].forward(xb, noiseless=True); ops["gaussian_blur"].transpose(yg)
    ym = ops["motion_blur"].forward(xb, noiseless=True); ops["motion_blur"].transpose(ym)
    ys = ops["super_resolution"].forward(xb, noiseless=True); ops["super_resolution"].forward_adjoint(ys); ops["super_resolution"].transpose(ys)
    yi, yif = ops["inpainting"].forward(xb, flatten=True); ops["inpainting"].transpose(yif, flatten=True)
    dwt.inv(dwt(xb))
run(); torch.cuda.synchronize()
L.check(lib.kdip_profile_enable(1))
for _ in range(3): run()
torch.cuda.synchronize()
dump = os.path.join(tempfile.gettempdir(), f"op_bw_{os.getpid()}.csv")
L.check(lib.kdip_profile_dump(dump.encode())); L.check(lib.kdip_profile_enable(0))
g = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(dump)):
    if r["class"].startswith("op_"):
        k = (r["class"], r["tag"], r["d0"], r["d1"], r["d2"], r["d3"]); g[k][0] += 1; g[k][1] += float(r["us"]); g[k][2] += float(r["mbytes"])
for k, (c, us, mb) in sorted(g.items()):
    print(f"{k}: {c} launches, {us / c:.1f} us, {mb / us * 1e3 if us else 0:.0f} GB/s")
