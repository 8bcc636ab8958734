"""Per-(kernel class, layer shape) time table of one guided Heun step (library HIP-event profiler).
usage: python tools/prof_shapes.py [batch] [step index]"""
import os, sys, csv, collections, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import kdip_amd._lib as L
import kdip_amd.unet as ku, kdip_amd.condition as kc, kdip_amd.measurements as km, kdip_amd.sampling as ks
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
step = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lib = L.load()
model = ku.UNetModel(dtype="bf16", **ku.FFHQ_CONFIG); model.load_state_dict(ku.synthetic_state_dict(seed=0, **ku.FFHQ_CONFIG))
D = ku.GaussianDiffusionTables()
op = km.get_operator("gaussian_blur", device="cuda", in_shape=(1, 3, 256, 256), kernel_size=61, intensity=3.0, sigma_s=0.05)
x0 = bench.smooth_image(B, 256, 1).cuda(); torch.manual_seed(2); meas = op.forward(x0.clone(), flatten=True)
den = kc.ConditionOpenAIDenoiser(inner_model=model, diffusion=D, x0_cov_type="convert", recon_mse=None, operator=op, measurement=meas, guidance="I", device="cuda")
sig = ks.get_sigmas_karras(100, 0.01, 80).cpu(); noise = torch.randn(B, 3, 256, 256, device="cuda")
x = x0 + float(sig[step]) * noise
for _ in range(2): ks.heun_step(den, x, sig, step)
torch.cuda.synchronize()
L.check(lib.kdip_profile_enable(1))
ks.heun_step(den, x, sig, step); torch.cuda.synchronize()
dump = os.path.join(tempfile.gettempdir(), "kdip_shapes.csv")
L.check(lib.kdip_profile_dump(dump.encode())); L.check(lib.kdip_profile_enable(0))
grp = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for r in csv.DictReader(open(dump)):
    k = (r["class"] + ("/small" if r["tag"] == "gn_small" else "/pool" if r["tag"] == "gn_pool" else ""), r["d0"], r["d1"], r["d2"], r["d3"])
    g = grp[k]; g[0] += 1; g[1] += float(r["us"]); g[2] += float(r["gflop"]); g[3] += float(r["mbytes"])
tot = sum(g[1] for g in grp.values())
print(f"profiled total {tot/1e3:.2f} ms (one Heun step = 2 guided calls)")
for k, g in sorted(grp.items(), key=lambda kv: -kv[1][1]):
    rate = f"{g[2]/g[1]*1e3:8.1f} TF/s" if g[2] > 0 else f"{g[3]/g[1]*1e3:8.1f} GB/s"   # MB/us = TB/s
    print(f"{k[0]:24s} {'x'.join(k[1:]):22s} n={g[0]:3d} avg {g[1]/g[0]:8.1f} us  tot {g[1]/1e3:7.2f} ms {100*g[1]/tot:5.1f}%  {rate}")
