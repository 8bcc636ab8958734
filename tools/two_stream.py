"""Experiment (usage: two_stream.py nsplit total_batch): a batch as nsplit part-batches on two HIP streams (fills the chip while one half is in
its latency-bound small-spatial layers) vs one batch-16 stream."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import kdip_amd.unet as ku, kdip_amd.condition as kc, kdip_amd.measurements as km, kdip_amd.sampling as ks
sd = ku.synthetic_state_dict(seed=0, **ku.FFHQ_CONFIG)
D = ku.GaussianDiffusionTables()
sig = ks.get_sigmas_karras(100, 0.01, 80).cpu()

def make(B, seed):
    model = ku.UNetModel(dtype="bf16", **ku.FFHQ_CONFIG); model.load_state_dict(sd)
    op = km.get_operator("gaussian_blur", device="cuda", in_shape=(1, 3, 256, 256), kernel_size=61, intensity=3.0, sigma_s=0.05)
    x0 = bench.smooth_image(B, 256, seed).cuda(); torch.manual_seed(2); meas = op.forward(x0.clone(), flatten=True)
    den = kc.ConditionOpenAIDenoiser(inner_model=model, diffusion=D, x0_cov_type="convert", recon_mse=None, operator=op, measurement=meas, guidance="I", device="cuda")
    return den, x0, torch.randn(B, 3, 256, 256, device="cuda")

nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 2
BT = int(sys.argv[2]) if len(sys.argv) > 2 else 16
parts = [make(BT // nsplit, 1 + k) for k in range(nsplit)]
streams = [torch.cuda.Stream() for _ in range(nsplit)]
def run(i, reps):
    for _ in range(reps):
        for (den, x0, nz), st in zip(parts, streams):
            with torch.cuda.stream(st):
                ks.heun_step(den, x0 + float(sig[i]) * nz, sig, i)
for i in (10, 95):
    run(i, 1); torch.cuda.synchronize()
    t = time.perf_counter(); run(i, 3); torch.cuda.synchronize()
    print(f"nsplit={nsplit} step {i}: {(time.perf_counter()-t)/3*1e3:.2f} ms per {BT}-image step")
