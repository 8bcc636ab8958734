"""DEV-ONLY: reference captures of the V2 denoiser with a transform-domain covariance (DWT-Var / DCT-Var), tests/golden/guided_calls_v2_ot.npz.

TEST INFRASTRUCTURE.  Build container only:  python -m oracle.make_golden_v2dwt

BASELINE configs[4] runs `ConditionOpenAIDenoiserV2` with `ortho_tf_type='dwt'` (condition/condition.py:277-300, :317-439 with
`ortho_tf`; condition/utils.py:106-163 `DiscreteWaveletTransform`, `LazyOTCovariance`).  Rounds 1 - 4 could not run that path of the
reference (PyWavelets absent from the interpreter the reference is imported with) and compared the HIP path with the oracle's own
restatement only.  Round 5 forwards the reference's four pywt calls to the real PyWavelets 1.1.1 in /opt/conda/bin/python3.9
(oracle/pywt_bridge.py), so the reference itself now produces these fixtures: tiny UNet + `out_cov` head, 64 x 64, three operators x
guidance I / II x sigma 1.5 (scalar variance), 0.5 and 0.12 (learned theta-variance: CG with the transform in the matvec), for the 'dwt'
and 'dct' bases.  (`autoI` needs GPyTorch and stays unpinned.)  Each capture is asserted equal to the oracle before it is written."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import refimport                                              # noqa: E402
from oracle import unet as ounet, operators as oops, condition as ocond   # noqa: E402
from oracle.make_golden import smooth_image, check, build_ref_model, GOLD  # noqa: E402


def main():
    ns = refimport.import_reference()
    cc, cm, ke = ns.cc, ns.cm, ns.ke
    import pywt
    assert getattr(pywt, "__kdip_bridge__", False), "the PyWavelets worker (/opt/conda/bin/python3.9) is not available"
    torch.set_num_threads(8)
    S = 64
    cfg = ounet.UNetConfig(**ounet.TINY)
    sd2 = ounet.init_state_dict(cfg, seed=0, out_cov=True)
    model, diffusion = build_ref_model(ns, ounet.TINY, sd2)
    x0 = smooth_image(1, S, seed=1)
    op_cfgs = {
        "gaussian_blur": dict(in_shape=(1, 3, S, S), kernel_size=61, intensity=3.0, sigma_s=0.05),
        "super_resolution": dict(in_shape=(1, 3, S, S), scale_factor=4, sigma_s=0.05),
        "inpainting": dict(sigma_s=0.05, mask_opt=dict(mask_type="random", mask_prob_range=(0.5, 0.5), image_size=S)),
    }
    dump = {}
    for basis in ("dwt", "dct"):
        den2 = ke.OpenAIDenoiserV2(model, diffusion, ortho_tf_type=basis)
        den2.out_cov = torch.nn.Conv2d(32, 6, 1)      # TINY has 32 feature channels; the reference hard-codes Conv2d(128, 6, 1) (external.py:141)
        den2.out_cov.weight.data.copy_(sd2["out_cov.weight"]); den2.out_cov.bias.data.copy_(sd2["out_cov.bias"])
        for name, kw in op_cfgs.items():
            with refimport.reference_cwd():
                np.random.seed(0)
                rop = cm.get_operator(name, device="cpu", **kw)
            np.random.seed(0)
            oop = oops.get_operator(name, **kw)
            torch.manual_seed(2)
            meas_r = rop.forward(x0.clone(), flatten=True)
            torch.manual_seed(2)
            meas_o = oop.forward(x0.clone(), flatten=True)
            for guidance in ("I", "II"):
                for sigma_v in (1.5, 0.5, 0.12):
                    x = x0 + sigma_v * torch.randn(1, 3, S, S, generator=torch.Generator().manual_seed(11))
                    sigma = torch.tensor([sigma_v])
                    rmodel = cc.ConditionOpenAIDenoiserV2(den2, operator=rop, measurement=meas_r, guidance=guidance, mle_sigma_thres=1.0,
                                                          device="cpu", ortho_tf_type=basis).eval()
                    omodel = ocond.GuidedDenoiser(sd2, cfg, oop, meas_o, guidance, mle_sigma_thres=1.0, v2=True, ortho_tf_type=basis)
                    h_r = rmodel(x.clone(), sigma)
                    h_o = omodel(x.clone(), sigma)
                    key = f"{name}|{guidance}|v2|{basis}|{sigma_v}"
                    check(key, h_r, h_o, 5e-4)
                    dump[key] = h_r.numpy()
    np.savez_compressed(os.path.join(GOLD, "guided_calls_v2_ot.npz"), **dump)
    print("wrote", len(dump), "captures")


if __name__ == "__main__":
    main()
