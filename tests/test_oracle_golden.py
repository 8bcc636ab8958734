"""CPU: the oracle restatement reproduces the committed golden vectors that
oracle/make_golden.py captured from the real reference."""
import numpy as np
import pytest
import torch

from oracle import tables as otab, unet as ounet, operators as oops
from oracle import transforms as otf, condition as ocond, sampling as osamp
from helpers import op_cfgs, synthetic_recon_mse

torch.set_num_threads(max(1, torch.get_num_threads()))


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_tables(gold):
    g = gold("tables")
    D = otab.DiffusionTables()
    assert torch.equal(otab.get_sigmas_karras(100, 0.01, 80), T(g["sigmas100"]))
    assert torch.equal(otab.get_sigmas_karras(20, 0.01, 80), T(g["sigmas20"]))
    t = D.sigma_to_t(T(g["probe"]))
    assert torch.equal(t, T(g["t_frac"]))
    assert torch.equal(t.long(), T(g["t_floor"]))
    assert np.array_equal(D.alphas_cumprod, g["alphas_cumprod"])
    assert torch.equal(otab.timestep_embedding(T(g["temb_t"]), 32), T(g["temb"]))
    # survey anchors: sigma -> floor(t)
    anchors = {80.0: 929, 10.0: 673, 1.0: 258, 0.2: 57, 0.05: 10, 0.01: 0}
    for s, tt in anchors.items():
        assert int(D.sigma_to_t(torch.tensor([s])).long()) == tt


def test_unet_forward_and_vjp(gold):
    g = gold("unet_tiny")
    cfg = ounet.UNetConfig(**ounet.TINY)
    sd = ounet.init_state_dict(cfg, seed=0)
    x = T(g["x"]).clone().requires_grad_()
    out, feat = ounet.unet_forward(sd, cfg, x, T(g["t"]), return_feature=True)
    assert float((out - T(g["out"])).abs().max()) < 1e-5
    assert float((feat - T(g["feature"])).abs().max()) < 1e-5
    v = torch.autograd.grad((out * T(g["cot"])).sum(), x)[0]
    assert float((v - T(g["vjp"])).abs().max()) < 1e-5


def test_unet_spec_counts():
    """FFHQ 12+3+12 blocks, 30 ResBlocks, 4 attention; ImageNet 18+3+18, 42, 16."""
    for cfgd, nin, nres, nattn in ((ounet.FFHQ, 12, 30, 4), (ounet.IMAGENET, 18, 42, 16)):
        cfg = ounet.UNetConfig(**cfgd)
        inp, mid, out, ch = ounet.unet_spec(cfg)
        layers = [L for b in inp + [mid] + out for L in b]
        assert len(inp) == nin and len(out) == nin
        assert sum(L[0] == "res" for L in layers) == nres
        assert sum(L[0] == "attn" for L in layers) == nattn
    shapes, _ = ounet.param_shapes(ounet.UNetConfig(**ounet.FFHQ))
    n = sum(int(np.prod(s)) for s in shapes.values())
    assert abs(n - 93.56e6) < 0.02e6


@pytest.mark.parametrize("name", ["gaussian_blur", "motion_blur", "super_resolution", "inpainting"])
def test_operators(gold, name):
    g = gold("operators")
    x0 = T(g["x0"])
    np.random.seed(0)
    op = oops.get_operator(name, **op_cfgs(64)[name])
    torch.manual_seed(2)
    y, yf = op.forward(x0.clone(), flatten=True)
    assert float((y - T(g[f"{name}.y"])).abs().max()) < 1e-6
    assert float((yf - T(g[f"{name}.y_flat"])).abs().max()) < 1e-6
    assert float((op.transpose(yf, flatten=True) - T(g[f"{name}.ATy"])).abs().max()) < 1e-6
    assert float((op.forward(x0.clone(), noiseless=True) - T(g[f"{name}.y_noiseless"])).abs().max()) < 1e-6
    if name == "inpainting":
        bits = np.unpackbits(g["inpainting.mask_bits"])[:64 * 64].reshape(64, 64)
        assert np.array_equal(op.mask[0, 0].numpy().astype(np.uint8), bits)


def test_mask_256_bit_exact(gold):
    g = gold("operators")
    np.random.seed(0)
    m = oops.random_mask(256, (0.5, 0.5))
    bits = np.unpackbits(g["inpainting.mask256_bits"]).reshape(256, 256)
    assert np.array_equal(m[0, 0].numpy().astype(np.uint8), bits)
    assert int(m[0, 0].sum()) == 65536 - 32768
    idx = torch.stack(torch.where(m > 0)[-3:])[:, :64]
    assert np.array_equal(idx.numpy(), g["inpainting.mask256_first_idx"])


def test_sr_resizer_256(gold):
    g = gold("operators")
    from helpers import smooth_image
    op = oops.get_operator("super_resolution", in_shape=(1, 3, 256, 256), scale_factor=4, sigma_s=0.05)
    y = op.forward(smooth_image(1, 256, 1), noiseless=True)
    assert y.shape == (1, 3, 64, 64)
    assert float((y - T(g["sr256.y_noiseless"])).abs().max()) < 1e-6


def test_transforms(gold):
    g = gold("transforms")
    x = T(g["x"])
    assert float((otf.dct_ortho(x) - T(g["dct"])).abs().max()) < 5e-6
    assert float((otf.idct_ortho(T(g["dct"])) - x).abs().max()) < 5e-6
    # Haar: definition-level self-checks (the pin against PyWavelets itself is tests/test_thirdparty_pins.py)
    w = otf.dwt_haar(x)
    assert float((otf.idwt_haar(w) - x).abs().max()) < 1e-5
    assert abs(float(w.norm() / x.norm()) - 1) < 1e-6
    c = torch.full((1, 1, 64, 64), 2.0)
    wc = otf.dwt_haar(c)
    assert abs(float(wc[0, 0, 0, 0]) - 16.0) < 1e-5          # cA3 = 2 * (sqrt2)^6
    assert float(wc[0, 0, 8:, :].abs().max()) == 0 and float(wc[0, 0, :8, 8:].abs().max()) == 0
    # a vertical edge between columns 0|1 excites 'ad' (approximation down the rows, detail along W) = the TOP-RIGHT block of level 1
    # (pywt.coeffs_to_array places blocks by key: first letter = rows, second = columns, 'd' = trailing half)
    e = torch.zeros(1, 1, 64, 64); e[..., :, 0] = 1
    we = otf.dwt_haar(e)
    assert float(we[0, 0, :32, 32:].abs().max()) > 0 and float(we[0, 0, 32:, :32].abs().max()) == 0


def _ops_and_meas(gold, name):
    g = gold("operators")
    np.random.seed(0)
    op = oops.get_operator(name, **op_cfgs(64)[name])
    torch.manual_seed(2)
    meas = op.forward(T(g["x0"]).clone(), flatten=True)
    return op, meas, T(g["x0"])


GUIDED = [("I", "convert", {}), ("II", "convert", {}), ("II", "pgdm", {}), ("dps", "dps", dict(zeta=1.0)),
          ("pgdm", "pgdm", {}), ("I", "analytic", {}), ("diffpir", "diffpir", dict(lambda_=7.0)),
          ("uncond", "convert", {}), ("dps+mle", "convert", dict(zeta=1.0))]


@pytest.mark.parametrize("name", ["gaussian_blur", "motion_blur", "super_resolution", "inpainting"])
def test_guided_calls(gold, name):
    g = gold("guided_calls")
    cfg = ounet.UNetConfig(**ounet.TINY)
    sd = ounet.init_state_dict(cfg, seed=0)
    op, meas, x0 = _ops_and_meas(gold, name)
    for guidance, cov, extra in GUIDED:
        for sigma_v in (1.5, 0.12):
            x = x0 + sigma_v * torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(11))
            m = ocond.GuidedDenoiser(sd, cfg, op, meas, guidance, x0_cov_type=cov, recon_mse=synthetic_recon_mse(),
                                     zeta=extra.get("zeta"), lambda_=extra.get("lambda_"))
            hat = m(x, torch.tensor([sigma_v]))
            ref = T(g[f"{name}|{guidance}|{cov}|{sigma_v}"])
            assert float((hat - ref).abs().max()) < 2e-4, (name, guidance, cov, sigma_v)


def test_guided_calls_v2(gold):
    g = gold("guided_calls_v2")
    cfg = ounet.UNetConfig(**ounet.TINY)
    sd = ounet.init_state_dict(cfg, seed=0, out_cov=True)
    for name in ("gaussian_blur", "inpainting", "super_resolution"):
        op, meas, x0 = _ops_and_meas(gold, name)
        for guidance in ("I", "II"):
            for sigma_v in (1.5, 0.12):
                x = x0 + sigma_v * torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(11))
                m = ocond.GuidedDenoiser(sd, cfg, op, meas, guidance, mle_sigma_thres=1.0, v2=True)
                hat = m(x, torch.tensor([sigma_v]))
                assert float((hat - T(g[f"{name}|{guidance}|v2|{sigma_v}"])).abs().max()) < 2e-4


def test_guided_calls_v2_transform_bases(gold):
    """Reference captures of the V2 denoiser with the DWT-Var / DCT-Var covariance (BASELINE configs[4]'s machinery: OrthoTransform inside
    the mat-solvers, condition/condition.py:317-439; condition/utils.py:88-163) -- produced by the REFERENCE with the real PyWavelets
    (oracle/make_golden_v2dwt.py, oracle/pywt_bridge.py): guidance I / II x sigma 1.5 (scalar variance), 0.5 and 0.12 (learned
    theta-variance, CG with the transform in the matvec)."""
    g = gold("guided_calls_v2_ot")
    cfg = ounet.UNetConfig(**ounet.TINY)
    sd = ounet.init_state_dict(cfg, seed=0, out_cov=True)
    for basis in ("dwt", "dct"):
        for name in ("gaussian_blur", "inpainting", "super_resolution"):
            op, meas, x0 = _ops_and_meas(gold, name)
            for guidance in ("I", "II"):
                for sigma_v in (1.5, 0.5, 0.12):
                    x = x0 + sigma_v * torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(11))
                    m = ocond.GuidedDenoiser(sd, cfg, op, meas, guidance, mle_sigma_thres=1.0, v2=True, ortho_tf_type=basis)
                    hat = m(x, torch.tensor([sigma_v]))
                    assert float((hat - T(g[f"{name}|{guidance}|v2|{basis}|{sigma_v}"])).abs().max()) < 2e-4, (basis, name, guidance, sigma_v)


def test_sampler_trajectories(gold):
    g = gold("sampler")
    cfg = ounet.UNetConfig(**ounet.TINY)
    sd = ounet.init_state_dict(cfg, seed=0)
    op, meas, _ = _ops_and_meas(gold, "gaussian_blur")
    sig = T(g["sigmas"])
    for sampler, fn in (("heun", osamp.sample_heun), ("euler", osamp.sample_euler)):
        m = ocond.GuidedDenoiser(sd, cfg, op, meas, "I", x0_cov_type="convert")
        x = fn(m, T(g["xT"]).clone(), sig)
        assert float((x - T(g[f"{sampler}.x0"])).abs().max()) < 5e-4
        torch.manual_seed(7)
        xc = fn(m, T(g["xT"]).clone(), sig, s_churn=80, s_tmin=0.05, s_tmax=50, s_noise=1.003)
        assert float((xc - T(g[f"{sampler}.x0_churn"])).abs().max()) < 5e-4


def test_error_behaviour():
    cfg = ounet.UNetConfig(**ounet.TINY)
    sd = ounet.init_state_dict(cfg, seed=0)
    np.random.seed(0)
    op = oops.get_operator("inpainting", **op_cfgs(64)["inpainting"])
    meas = op.forward(torch.zeros(1, 3, 64, 64), flatten=True)
    with pytest.raises(ValueError):
        ocond.GuidedDenoiser(sd, cfg, op, meas, "bogus")(torch.zeros(1, 3, 64, 64), torch.tensor([1.0]))
    with pytest.raises(AssertionError):
        ocond.GuidedDenoiser(sd, cfg, op, meas, "dps", x0_cov_type="dps")(torch.zeros(1, 3, 64, 64), torch.tensor([1.0]))


def test_extra_guidance_and_analytic(gold):
    """tmpd covariance, STSL guidance and the analytic-variance estimator against the reference captures."""
    from oracle import analytic as oana
    from helpers import smooth_image
    g = gold("guided_calls_extra")
    cfg = ounet.UNetConfig(**ounet.TINY)
    sd = ounet.init_state_dict(cfg, seed=0)
    modes = [("I", "tmpd", {}), ("II", "tmpd", {}), ("stsl", "dps", dict(zeta=1.0, eta=0.5, num_hutchinson_samples=2)),
             ("stsl+mle", "convert", dict(zeta=1.0, eta=0.5, num_hutchinson_samples=2))]
    for name in ("gaussian_blur", "inpainting"):
        op, meas, x0 = _ops_and_meas(gold, name)
        for guidance, cov, extra in modes:
            for sigma_v in (1.5, 0.12):
                key = f"{name}|{guidance}|{cov}|{sigma_v}"
                if key not in g.files:
                    continue
                x = x0 + sigma_v * torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(11))
                m = ocond.GuidedDenoiser(sd, cfg, op, meas, guidance, x0_cov_type=cov, zeta=extra.get("zeta"), eta=extra.get("eta"),
                                         num_hutchinson_samples=extra.get("num_hutchinson_samples"))
                torch.manual_seed(5)
                assert float((m(x, torch.tensor([sigma_v])) - T(g[key])).abs().max()) < 5e-4, key
    ga = gold("analytic_variance")
    torch.manual_seed(9)
    est = oana.estimate_recon_mse(sd, cfg, [smooth_image(2, 64, 21), smooth_image(2, 64, 22)], T(ga["sigmas"]))
    assert float((est["mse_list"] - T(ga["mse_list"])).abs().max()) < 1e-5


def test_sampler_dpmpp2m_trajectory(gold):
    """oracle sample_dpmpp_2m (sampling.py:583-605) against the reference capture (5 steps, Type-I + Convert, tiny model)."""
    g = gold("sampler_dpmpp2m")
    cfg = ounet.UNetConfig(**ounet.TINY)
    sd = ounet.init_state_dict(cfg, seed=0)
    op, meas, _ = _ops_and_meas(gold, "gaussian_blur")
    assert torch.equal(meas[0], T(g["y"]))                 # same measurement as the operator fixtures
    first = []
    m = ocond.GuidedDenoiser(sd, cfg, op, meas, "I", x0_cov_type="convert")
    x = osamp.sample_dpmpp_2m(m, T(g["xT"]).clone(), T(g["sigmas"]), callback=lambda d: first.append(d["denoised"]))
    assert float((first[0] - T(g["denoised_first"])).abs().max()) < 5e-4
    assert float((x - T(g["x0"])).abs().max()) < 5e-4


def test_box_masks_bit_exact(gold):
    """box / extreme inpainting masks (measurements.py:264-320): numpy seed -> identical bits, oracle and host generator."""
    import numpy as np
    import kdip_amd.measurements as km
    from oracle import operators as oops
    g = gold("masks_box")
    assert len(g.files) == 6
    for key in g.files:
        mt, seed, S, lo, hi = key.split("|")
        seed, S, lo, hi = int(seed), int(S), int(lo), int(hi)
        want = np.unpackbits(g[key]).reshape(S, S).astype(np.float32)
        np.random.seed(seed)
        o = oops.box_mask(S, (lo, hi), extreme=(mt == "extreme"))
        assert np.array_equal(o[0, 0].numpy(), want) and np.array_equal(o[0, 2].numpy(), want), key
        np.random.seed(seed)
        h = km.MaskGenerator(mask_type=mt, mask_len_range=(lo, hi), image_size=S)(torch.empty(1, 3, S, S))
        assert h.shape == (1, 3, S, S) and np.array_equal(h[0, 1].numpy(), want), key
