"""Timing of guided Heun steps on the DWT-Var path (BASELINE configs[5] shape: FFHQ + out_cov head, Gaussian deblur, autoI,
ortho_tf dwt, mle_sigma_thres 1 -> CG with the Haar DWT inside the matvec below sigma 1).  usage: python tools/v2_call.py [batch]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import kdip_amd.unet as ku, kdip_amd.condition as kc, kdip_amd.measurements as km, kdip_amd.sampling as ks
from kdip_amd.external import OpenAIDenoiserV2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model = ku.UNetModel(dtype="bf16", **ku.FFHQ_CONFIG); model.load_state_dict(ku.synthetic_state_dict(seed=0, out_cov=True, **ku.FFHQ_CONFIG))
D = ku.GaussianDiffusionTables()
op = km.get_operator("gaussian_blur", device="cuda", in_shape=(1, 3, 256, 256), kernel_size=61, intensity=3.0, sigma_s=0.05)
x0 = bench.smooth_image(B, 256, 1).cuda(); torch.manual_seed(2); meas = op.forward(x0.clone(), flatten=True)
den2 = OpenAIDenoiserV2(model, D, device="cuda", ortho_tf_type="dwt")
den = kc.ConditionOpenAIDenoiserV2(denoiser=den2, operator=op, measurement=meas, guidance="autoI", device="cuda", mle_sigma_thres=1.0, ortho_tf_type="dwt")
sig = ks.get_sigmas_karras(100, 0.01, 80).cpu(); noise = torch.randn(B, 3, 256, 256, device="cuda")
for i in (10, 70, 85, 95):
    x = x0 + float(sig[i]) * noise
    ks.heun_step(den, x, sig, i); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(2): ks.heun_step(den, x, sig, i)
    torch.cuda.synchronize()
    print(f"V2/DWT step {i} sigma {float(sig[i]):.3f}: {(time.perf_counter() - t) / 2 * 1e3:.1f} ms per Heun step at B={B}  cg_iters(max) {max(getattr(op, 'cg_iters', [0]) or [0])}")
