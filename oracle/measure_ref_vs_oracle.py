"""DEV-ONLY: seconds per full-size guided call of the REAL reference vs the oracle restatement at equal core count (SURVEY.md 8d asks
for agreement within 10 %): FFHQ architecture, 256 x 256, Gaussian deblur, Type-I + Convert, batch 1, one closed-form call
(sigma = 47) and one CG-branch call (sigma = 0.05), best of 3 after a warm-up, torch threads = the cores of this container.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference).  Usage: python -m oracle.measure_ref_vs_oracle
Writes profiles/r04/ref_vs_oracle_cpu.json -- bench.py's cpu_baseline leg quotes it as `reference_cross_check`."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refimport, unet as ounet, operators as oops, condition as ocond     # noqa: E402
from oracle.make_golden import build_ref_model, smooth_image                           # noqa: E402


def best_of(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    ncores = len(os.sched_getaffinity(0))
    torch.set_num_threads(ncores)
    ns = refimport.import_reference()
    cc, cm = ns.cc, ns.cm
    cfg = ounet.UNetConfig(**ounet.FFHQ)
    sd = ounet.init_state_dict(cfg, seed=0)
    model, diffusion = build_ref_model(ns, ounet.FFHQ, sd)
    x0 = smooth_image(1, 256, 1)
    kw = dict(in_shape=(1, 3, 256, 256), kernel_size=61, intensity=3.0, sigma_s=0.05)
    with refimport.reference_cwd():
        rop = cm.get_operator("gaussian_blur", device="cpu", **kw)
    oop = oops.get_operator("gaussian_blur", **kw)
    torch.manual_seed(2); meas_r = rop.forward(x0.clone(), flatten=True)
    torch.manual_seed(2); meas_o = oop.forward(x0.clone(), flatten=True)
    rmodel = cc.ConditionOpenAIDenoiser(inner_model=model, diffusion=diffusion, x0_cov_type="convert", recon_mse=None, operator=rop,
                                        measurement=meas_r, guidance="I", mle_sigma_thres=0.2, device="cpu").eval()
    omodel = ocond.GuidedDenoiser(sd, cfg, oop, meas_o, "I", x0_cov_type="convert")
    out = {"cores": ncores, "torch": torch.__version__, "workload": "FFHQ 256x256 Gaussian deblur, Type-I + Convert, batch 1, one guided call", "calls": {}}
    for tag, sigma_v in (("closed_form_sigma_47", 47.0), ("cg_branch_sigma_0.05", 0.05)):
        x = x0 + sigma_v * torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(3))
        s = torch.tensor([sigma_v])
        h_r = rmodel(x.clone(), s); h_o = omodel(x.clone(), s)
        tr = best_of(lambda: rmodel(x.clone(), s))
        to = best_of(lambda: omodel(x.clone(), s))
        out["calls"][tag] = {"reference_s_per_call": round(tr, 4), "oracle_s_per_call": round(to, 4), "oracle_over_reference": round(to / tr, 4),
                             "max_abs_reference_minus_oracle": float((h_r - h_o).abs().max())}
        print(tag, out["calls"][tag])
    dst = os.path.join(ROOT, "profiles", "r04", "ref_vs_oracle_cpu.json")
    json.dump(out, open(dst, "w"), indent=1)
    print("written", dst)


if __name__ == "__main__":
    main()
