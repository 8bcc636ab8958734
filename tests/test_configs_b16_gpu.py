"""GPU: every BASELINE.json config at its PER-GPU batch on one MI355X (VERDICT r5 missing #2 / weak #4).  The 2- / 4- / 8-GPU
configs are contiguous 16-image shards of these workloads (k_diffusion/evaluation.py:53-63; configs/test_imagenet.json:13-17),
so one guided call at batch 16 (configs[0]: batch 1) in the headline arithmetic (bf16x3) is exactly one rank's per-call work:
it must fit, be finite, in range, and -- the parity modes are deterministic -- repeat bit for bit.  The workspace the call needs
is printed.  Built through bench.build_problem, i.e. the same objects `bench.py --workload` times."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("wl", ["cfg0", "cfg1", "cfg2", "cfg3", "cfg4"])
def test_config_runs_at_per_gpu_batch(wl):
    import bench
    import kdip_amd.unet as ku
    import kdip_amd.sampling as ks
    WL = bench.WORKLOADS[wl]
    B = WL["batch"]
    assert B == (1 if wl == "cfg0" else 16)
    dev = torch.device("cuda", 0)
    arch = ku.FFHQ_CONFIG if WL["arch"] == "FFHQ" else ku.IMAGENET_CONFIG
    sd = ku.synthetic_state_dict(seed=0, out_cov=bool(WL.get("ortho")), **arch)
    D = ku.GaussianDiffusionTables()
    den, op, x0, meas = bench.build_problem(WL, "bf16x3", dev, B, sd, D, seed=0)
    sig = ks.get_sigmas_karras(WL["nsteps"], 0.01, 80, rho=7.0, device="cpu")
    noise = torch.randn(B, 3, 256, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    torch.cuda.reset_peak_memory_stats()
    for i in (WL["nsteps"] // 10, WL["nsteps"] * 95 // 100):          # one high-sigma (closed form) and one low-sigma (CG where the config has one) call
        s = float(sig[i])
        x = (x0 + s * noise).contiguous()
        sv = torch.full((B,), s, device=dev)
        a = den(x, sv).clone()
        b = den(x, sv).clone()
        torch.cuda.synchronize()
        assert a.shape == (B, 3, 256, 256) and torch.isfinite(a).all(), (wl, s)
        assert float(a.abs().max()) <= 1.0 + 1e-6, (wl, s, float(a.abs().max()))      # guided outputs are clipped to [-1, 1] (condition.py:131,173)
        assert torch.equal(a, b), (wl, s, float((a - b).abs().max()))                 # deterministic parity mode: bitwise repeatable
    u = bench.unet_of(den)
    assert u.x3_saturated() == 0
    if wl == "cfg1":      # the opt-in fp16-headed split on the benchmarked config at its batch: in-window (no call redone bf16-headed), repeatable
        del den
        torch.cuda.empty_cache()
        denh, _, _, _ = bench.build_problem(WL, "f16x3", dev, B, sd, D, seed=0)
        s = float(sig[10])
        x = (x0 + s * noise).contiguous()
        a = denh(x, torch.full((B,), s, device=dev)).clone()
        b = denh(x, torch.full((B,), s, device=dev)).clone()
        assert torch.isfinite(a).all() and torch.equal(a, b) and bench.unet_of(denh).x3_fallbacks == 0
    print(f"\n{wl} B={B} bf16x3: UNet workspace {u.workspace_bytes(B) / 1e9:.2f} GB, torch peak {torch.cuda.max_memory_allocated() / 1e9:.2f} GB ({WL['label'].split(':')[0]})")
