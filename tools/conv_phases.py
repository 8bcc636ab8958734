"""Per-block phase times of one conv layer shape inside a real guided Heun step (needs a -DKDIP_TIMING=1 build:
python k-diffusion-inverse-problems_amd/build.py --variant timing -DKDIP_TIMING=1; KDIP_LIB_PATH=...libkdip_hip_timing.so).
usage: python tools/conv_phases.py H cin cout st_mode [batch] [dtype bf16|bf16x3|f32]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import kdip_amd._lib as L
import kdip_amd.unet as ku, kdip_amd.condition as kc, kdip_amd.measurements as km, kdip_amd.sampling as ks
H, cin, cout, mode = [int(a) for a in sys.argv[1:5]]
B = int(sys.argv[5]) if len(sys.argv) > 5 else 16
DT = sys.argv[6] if len(sys.argv) > 6 else "bf16"
lib = L.load()
model = ku.UNetModel(dtype=DT, **ku.FFHQ_CONFIG); model.load_state_dict(ku.synthetic_state_dict(seed=0, **ku.FFHQ_CONFIG))
D = ku.GaussianDiffusionTables()
op = km.get_operator("gaussian_blur", device="cuda", in_shape=(1, 3, 256, 256), kernel_size=61, intensity=3.0, sigma_s=0.05)
x0 = bench.smooth_image(B, 256, 1).cuda(); torch.manual_seed(2); meas = op.forward(x0.clone(), flatten=True)
den = kc.ConditionOpenAIDenoiser(inner_model=model, diffusion=D, x0_cov_type="convert", recon_mse=None, operator=op, measurement=meas, guidance="I", device="cuda")
sig = ks.get_sigmas_karras(100, 0.01, 80).cpu(); noise = torch.randn(B, 3, 256, 256, device="cuda")
x = x0 + float(sig[10]) * noise
ks.heun_step(den, x, sig, 10); torch.cuda.synchronize()
buf = torch.zeros(16384 * 8, dtype=torch.int64, device="cuda")
L.check(lib.kdip_debug_conv_timing(L.ptr(buf), H, cin, cout, mode))
ks.heun_step(den, x, sig, 10); torch.cuda.synchronize()
L.check(lib.kdip_debug_conv_timing(None, 0, 0, 0, 0))
t = buf.cpu().view(-1, 8).double()
t = t[t[:, 0] > 0]
n = t.shape[0]
if n == 0:
    print("no matching launch"); sys.exit(0)
t0 = t[:, 0].min()
us = lambda v: float(v) / 100.0          # 100 MHz ticks -> us
print(f"blocks {n}; kernel span {us(t[:, 3].max() - t0):.1f} us")
pro, kl, ep, life = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 3] - t[:, 0]
rows = [("prologue (index math + first patch)", pro), ("K loop", kl), ("epilogue", ep), ("  epilogue: transpose + store loop", t[:, 4] - t[:, 2])]
if float(t[:, 5].max()) > 0:
    rows += [("  epilogue: stats shuffles + barrier", t[:, 5] - t[:, 4]), ("  epilogue: LDS atomics + barrier", t[:, 6] - t[:, 5]), ("  epilogue: global atomics", t[:, 3] - t[:, 6])]
rows.append(("block lifetime", life))
for name, v in rows:
    print(f"  {name:38s} mean {us(v.mean()):7.2f} us  p10 {us(v.quantile(0.1)):7.2f}  p50 {us(v.quantile(0.5)):7.2f}  p90 {us(v.quantile(0.9)):7.2f}")
# residency: average number of concurrently live blocks
ev = torch.cat([torch.stack([t[:, 0], torch.ones(n, dtype=torch.double)], 1), torch.stack([t[:, 3], -torch.ones(n, dtype=torch.double)], 1)])
ev = ev[ev[:, 0].argsort()]
live = ev[:, 1].cumsum(0)
dt = ev[1:, 0] - ev[:-1, 0]
print(f"  mean live blocks {float((live[:-1] * dt).sum() / dt.sum()):.0f} (768 = 3 per CU)")
