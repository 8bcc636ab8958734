"""One guided Heun step of the bench workload (BASELINE configs[1], one part-batch of 8 images, one stream) with the library's
launch profiler on from the very first launch, so that the n-th conv3 record of the dump is the n-th conv3 dispatch of the
process -- tools/pmc_join.py joins it with a `rocprofv3 --pmc` pass of this same command.  usage: python tools/pmc_step.py dump.csv [batch] [dtype]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kdip_amd._lib as L
import kdip_amd.unet as ku, kdip_amd.condition as kc, kdip_amd.measurements as km, kdip_amd.sampling as ks
from bench import smooth_image
dump = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
DT = sys.argv[3] if len(sys.argv) > 3 else os.environ.get("PMC_DTYPE", "bf16")
lib = L.load()
L.check(lib.kdip_profile_enable(1))
dev = "cuda"
D = ku.GaussianDiffusionTables()
sig = ks.get_sigmas_karras(100, 0.01, 80, rho=7.0, device=dev).cpu()
model = ku.UNetModel(dtype=DT, device=dev, **ku.FFHQ_CONFIG)
model.load_state_dict(ku.synthetic_state_dict(seed=0, **ku.FFHQ_CONFIG))
op = km.get_operator("gaussian_blur", device=dev, in_shape=(1, 3, 256, 256), kernel_size=61, intensity=3.0, sigma_s=0.05)
x0 = smooth_image(B, 256, 1).to(dev)
torch.manual_seed(2)
meas = op.forward(x0.clone(), flatten=True)
den = kc.ConditionOpenAIDenoiser(inner_model=model, diffusion=D, x0_cov_type="convert", recon_mse=None, operator=op, measurement=meas,
                                 guidance="I", mle_sigma_thres=0.2, device=dev).eval()
noise = torch.randn(B, 3, 256, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
for i in (10, 10):            # the first step also sizes the workspaces; both are recorded and joined
    x = (x0 + float(sig[i]) * noise).contiguous()
    ks.heun_step(den, x, sig, i)
torch.cuda.synchronize()
L.check(lib.kdip_profile_dump(dump.encode()))
print("dumped", dump)
