"""Timing of one guided call (Type-I + analytic covariance, motion deblur = BASELINE configs[3] shape) on the ImageNet-256 UNet.
usage: python tools/imagenet_call.py [batch] [dtype bf16 | bf16x3 | f32]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import kdip_amd.unet as ku, kdip_amd.condition as kc, kdip_amd.measurements as km, kdip_amd.sampling as ks
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
model = ku.UNetModel(dtype=dtype, **ku.IMAGENET_CONFIG); model.load_state_dict(ku.synthetic_state_dict(seed=0, **ku.IMAGENET_CONFIG))
D = ku.GaussianDiffusionTables()
op = km.get_operator("motion_blur", device="cuda", in_shape=(1, 3, 256, 256), kernel_size=61, intensity=0.5, sigma_s=0.05)
x0 = bench.smooth_image(B, 256, 1).cuda(); torch.manual_seed(2); meas = op.forward(x0.clone(), flatten=True)
s = ks.get_sigmas_karras(1000, 0.01, 80, device="cpu")[:-1]
recon = {"sigmas": s, "mse_list": s ** 2 / (1 + s ** 2) * 0.5}
den = kc.ConditionOpenAIDenoiser(inner_model=model, diffusion=D, x0_cov_type="analytic", recon_mse=recon, operator=op, measurement=meas, guidance="I", device="cuda")
sig = ks.get_sigmas_karras(100, 0.01, 80).cpu(); noise = torch.randn(B, 3, 256, 256, device="cuda")
for i in (10, 95):
    x = x0 + float(sig[i]) * noise
    ks.heun_step(den, x, sig, i); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(2): ks.heun_step(den, x, sig, i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 2 * 1e3
    print(f"ImageNet {dtype} step {i}: {ms:.1f} ms per Heun step at B={B} -> {2 * B * 4491.40 / ms:.0f} TFLOP/s algorithmic, {B / (ms * 100 / 1e3):.3f} images/s")
