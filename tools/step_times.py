"""Per-step wall time at a closed-form step (i=10) and a CG step (i=95) + CG iteration counts."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import kdip_amd.unet as ku, kdip_amd.condition as kc, kdip_amd.measurements as km, kdip_amd.sampling as ks
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model = ku.UNetModel(dtype="bf16", **ku.FFHQ_CONFIG); model.load_state_dict(ku.synthetic_state_dict(seed=0, **ku.FFHQ_CONFIG))
D = ku.GaussianDiffusionTables()
op = km.get_operator("gaussian_blur", device="cuda", in_shape=(1, 3, 256, 256), kernel_size=61, intensity=3.0, sigma_s=0.05)
x0 = bench.smooth_image(B, 256, 1).cuda(); torch.manual_seed(2); meas = op.forward(x0.clone(), flatten=True)
den = kc.ConditionOpenAIDenoiser(inner_model=model, diffusion=D, x0_cov_type="convert", recon_mse=None, operator=op, measurement=meas, guidance="I", device="cuda")
sig = ks.get_sigmas_karras(100, 0.01, 80).cpu(); noise = torch.randn(B, 3, 256, 256, device="cuda")
for i in (10, 95, 10, 95, 85, 99):
    x = x0 + float(sig[i]) * noise
    ks.heun_step(den, x, sig, i); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): ks.heun_step(den, x, sig, i)
    torch.cuda.synchronize()
    print(f"step {i} sigma {float(sig[i]):.4f}: {(time.perf_counter()-t)/3*1e3:.2f} ms  cg_iters {getattr(op,'cg_iters',None)}")
