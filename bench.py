#!/usr/bin/env python
"""Benchmark of the hot path: BASELINE.json configs[1] -- FFHQ 256x256 Gaussian deblur, Type-I
guidance with Convert posterior covariance, 100 Heun steps, batch 16 per MI355X (--batch) run as
--streams (2) independent part-batches on their own HIP streams / host threads.  The headline runs in the f16x3 arithmetic
(fp32 storage, split-precision convs -- three fp16 MFMAs per product, 11 + 11-bit operands --, deterministic reductions, every guided call
polled by a two-sided fp16-window watch and redone bf16-headed when it fires: `x3_fallbacks` on the line): the fast mode that meets
north_star's 1e-3 dB against the reference's fp32 arithmetic (conv error 4.7e-7 vs fp64, every parity test at the f32 / bf16x3 bounds);
the bf16-headed split (bf16x3, the headline of rounds 5 -- 6a), the bf16 throughput mode and the exact-f32 mode are carried as extra legs.  At N = 1 the same run is
repeated at batch 128 and reported as `throughput_at_batch_128` (per-image cost falls with batch).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: either under a launcher -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ... -- or
   plainly as `python bench.py --gpus N`, which starts the N ranks itself)

A "step" is one Heun sampler step of the 100-step Karras schedule over one batch of 16 synthetic
images (2 guided-denoiser calls = 2 UNet forwards + 2 hand-written UNet VJPs + 2 mat-solves, CG
on the sigma < 0.2 steps).  With K = 100 (default) the timed region is the whole sampler run; with
K < 100 the K timed steps are spread evenly over the schedule (so the closed-form / CG mix is
preserved) and each starts from x0 + sigma_i * noise.  Inputs are resident in HBM before the
timed region.  value = images/s of the whole job = N * batch / (100 * seconds_per_step).

Extra objects on the JSON line:
  roofline     dominant kernel (bf16 3x3 implicit-GEMM conv): algorithmic FLOPs / HIP-event time,
               measured live in an extra profiled pass after the timed region
               (+ hbm_bound_classes: GroupNorm / operator / point-wise kernels in GB/s; op_bandwidth_at_64_images)
  cpu_baseline the CPU oracle (torch-CPU fp32 restatement of the reference, oracle/) timed on the
               host cores on a bounded sample (rank 0, N = 1 only); reference_cross_check = the imported
               reference vs the oracle at equal cores (profiles/r04/ref_vs_oracle_cpu.json)
  bf16_throughput_mode / f32_parity_mode   the same workload in the other two arithmetic modes (bf16: faster, narrower than the
               reference, NOT the headline; f32: exact-f32 MFMA)
  hipgraph_replay, throughput_at_batch_128   extra legs (N = 1)
  ranks, distinct_devices, backend, gather_ms   multi-GPU evidence (one process per GPU, RCCL all_gather at the end)
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_MFMA_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
CALLS_PER_IMAGE = 199               # 100 Heun steps, last one Euler
FWD_VJP_GFLOP_PER_IMAGE_CALL = 776.26   # FFHQ UNet forward + input-VJP, 2*MAC

# ---- the five BASELINE.json configs as single-GPU workloads (`--workload`; the default cfg1 is the one `metric` is quoted on).  The
# multi-GPU configs are contiguous 16-image shards of these (condition/condition.py:84 asserts batch 1 in the reference: a batch is B
# independent problems), so one GPU at the per-GPU batch IS the per-rank work of configs[2] (128 / 8), [3] (64 / 4) and [4] (32 / 2).
# gflop_call: algorithmic UNet work per image per guided call (2*MAC; forward + input-VJP, forward only for Type II).
WORKLOADS = {
    "cfg0": dict(label="BASELINE configs[0]: FFHQ 256x256 inpainting (random mask, p=0.5), DPS guidance (zeta=1), 20 Euler steps, batch 1",
                 ref="condition/condition.py:140-148, condition/measurements.py:202-244, k_diffusion/sampling.py:118-135",
                 arch="FFHQ", op="inpainting", guidance="dps", cov="dps", zeta=1.0, sampler="euler", nsteps=20, batch=1, gflop_call=776.26),
    "cfg1": dict(label="BASELINE configs[1]: FFHQ 256x256 Gaussian deblur (61x61 PSF, sigma_s=0.05), Type-I guidance, Convert covariance (CG below sigma 0.2), 100 Heun steps (--ode)",
                 ref="condition/condition.py:167-174,231-274", arch="FFHQ", op="gaussian_blur", guidance="I", cov="convert", sampler="heun", nsteps=100, batch=16,
                 gflop_call=776.26),
    "cfg2": dict(label="BASELINE configs[2]: FFHQ 256x256 4x super-resolution (bicubic Resizer), Type-II guidance, PiGDM covariance, 100 Heun steps; 16 images = one rank's shard of the batch-128 / 8-GPU job",
                 ref="condition/condition.py:176-183, condition/measurements.py:86-122", arch="FFHQ", op="super_resolution", guidance="II", cov="pgdm", sampler="heun",
                 nsteps=100, batch=16, gflop_call=387.93),
    "cfg3": dict(label="BASELINE configs[3]: ImageNet-256 architecture (256 channels, 2 ResBlocks per level, attention at 32/16/8), motion deblur (61x61), Type-I guidance, Analytic covariance, 100 Heun steps; 16 images = one rank's shard of the batch-64 / 4-GPU job",
                 ref="configs/test_imagenet.json:13-17, condition/condition.py:250-254", arch="IMAGENET", op="motion_blur", guidance="I", cov="analytic", sampler="heun",
                 nsteps=100, batch=16, gflop_call=4491.40),
    "cfg4": dict(label="BASELINE configs[4]: FFHQ 256x256 Gaussian deblur, DWT-Var covariance (out_cov head, Haar-3 basis), auto Type-I guidance (= Type-I gradient with the CG solve in the DWT basis below sigma 1), 100 Heun steps; 16 images = one rank's shard of the batch-32 / 2-GPU job",
                 ref="condition/condition.py:133-138,287-300, k_diffusion/external.py:161-169", arch="FFHQ", op="gaussian_blur", guidance="autoI", cov=None, ortho="dwt", sampler="heun",
                 nsteps=100, batch=16, gflop_call=776.26),
}
OPKW = {"gaussian_blur": dict(kernel_size=61, intensity=3.0, sigma_s=0.05), "motion_blur": dict(kernel_size=61, intensity=0.5, sigma_s=0.05),
        "super_resolution": dict(scale_factor=4, sigma_s=0.05), "inpainting": dict(sigma_s=0.05)}


def smooth_image(B, size, seed):
    g = torch.Generator().manual_seed(seed)
    r = torch.rand(B, 3, size, size, generator=g) * 2 - 1
    rp = torch.nn.functional.pad(r, (4, 4, 4, 4), mode="circular")
    return (3 * torch.nn.functional.avg_pool2d(rp, 9, 1)).clamp(-1, 1)


def step_indices(K, n=100, last_single=True):
    """K >= 100: whole sampler runs.  K < 100: K two-call Heun steps spread evenly over steps 0..98 -- the single-call final
    Euler step (step 99) is never part of a subset, so a subset can only read slower per step than the full run (by 0.5 %)."""
    if K >= n:
        return list(range(n)) * (K // n) + list(range(K % n))
    if K == 1:
        return [n // 2]
    return [int(round(j * (n - (2 if last_single else 1)) / (K - 1))) for j in range(K)]


def available_cores():
    """Host cores this process may actually use: min(affinity mask, cgroup v2/v1 CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def build_problem(WL, dtype, dev, Bk, sd, D, seed=0, S=256):
    """One part-batch of a workload: UNet handle + operator + synthetic measurement + guided denoiser, exactly as the harness builds them
    (sample_condition_openai.py:71-217 / sample_condition_openai_v2.py:64-198).  Returns (denoiser, operator, x0, measurement)."""
    import kdip_amd.unet as ku
    import kdip_amd.condition as kc
    import kdip_amd.measurements as km
    import kdip_amd.sampling as ks
    ARCH = ku.FFHQ_CONFIG if WL["arch"] == "FFHQ" else ku.IMAGENET_CONFIG
    model = ku.UNetModel(dtype=dtype, device=dev, **ARCH)
    model.load_state_dict(sd)
    # fp16 window of the VJP's gradient operands.  Default: one power-of-two scale per VJP (from max |cotangent|).  The ImageNet-256 architecture's high-sigma
    # gradients span more than that window holds (12 - 15 dgrad launches per call at 5e-5 ... 2e-4 of the cotangent): in f16x3 every such call would be flagged and
    # redone bf16-headed (623 ms per step), so this workload lets every dgrad launch scale by a sampled maximum of its own input (deterministic: a function of the
    # data; 453 vs 475 ms for bf16x3, no call flagged -- profiles/r06/ab_cfg3_window.log).  KDIP_BENCH_X3_WINDOW=vjp|launch overrides.
    win = os.environ.get("KDIP_BENCH_X3_WINDOW", "launch" if (WL["arch"] == "IMAGENET" and dtype == "f16x3") else "vjp")
    if dtype in ("f16x3", "bf16x3") and win == "launch":
        model.set_x3_window("launch")
    # operator + synthetic measurement (sigma_s = 0.05), rank- and part-offset seeds: every image is its own problem
    opkw = dict(OPKW[WL["op"]])
    if WL["op"] == "inpainting":
        np.random.seed(seed)                                  # the mask generator draws from numpy's global stream (measurements.py:264-320)
        opkw["mask_opt"] = dict(mask_type="random", mask_prob_range=(0.5, 0.5), image_size=S)
    else:
        opkw["in_shape"] = (1, 3, S, S)
    op = km.get_operator(WL["op"], device=dev, **opkw)
    x0 = smooth_image(Bk, S, seed=1 + seed).to(dev)
    torch.manual_seed(2 + seed)
    meas = op.forward(x0.clone(), flatten=True)
    if WL.get("ortho"):      # V2 denoiser: out_cov head + covariance in the transform basis
        from kdip_amd.external import OpenAIDenoiserV2
        den = kc.ConditionOpenAIDenoiserV2(OpenAIDenoiserV2(model, D, device=dev, ortho_tf_type=WL["ortho"]), operator=op, measurement=meas,
                                           guidance=WL["guidance"], device=dev, mle_sigma_thres=1.0, ortho_tf_type=WL["ortho"]).eval()
    else:
        recon = None
        if WL["cov"] == "analytic":      # the analytic-variance table (analytic_variance.py:113-139) is an input of the sampler: synthetic, like the weights
            s_ = ks.get_sigmas_karras(1000, 0.01, 80, device="cpu")[:-1]
            recon = {"sigmas": s_.to(dev), "mse_list": (s_ ** 2 / (1 + s_ ** 2) * 0.5).to(dev)}
        den = kc.ConditionOpenAIDenoiser(inner_model=model, diffusion=D, x0_cov_type=WL["cov"], recon_mse=recon, operator=op,
                                         measurement=meas, guidance=WL["guidance"], zeta=WL.get("zeta"), mle_sigma_thres=0.2, device=dev).eval()
    return den, op, x0, meas


def unet_of(den):
    return den.inner_model if hasattr(den, "inner_model") else den.denoiser.inner_model


REF_VS_ORACLE = os.path.join(ROOT, "profiles", "r04", "ref_vs_oracle_cpu.json")     # oracle/measure_ref_vs_oracle.py (build container)


def cpu_baseline(sig):
    """Oracle (kind 'port') timed on the host: 3 Heun steps (6 Type-I/Convert guided calls) of the real schedule in the
    closed-form regime (steps 10-12) and 3 in the CG regime (steps 95-97) at batch 1 through the oracle's own sample_heun,
    extrapolated with the schedule's 157 closed-form + 42 CG calls per image."""
    from oracle import unet as ounet, operators as oops, condition as ocond, sampling as osamp
    ncores = available_cores()
    torch.set_num_threads(ncores)
    cfg = ounet.UNetConfig(**ounet.FFHQ)
    sd = ounet.init_state_dict(cfg, seed=0)
    x0 = smooth_image(1, 256, 1)
    op = oops.get_operator("gaussian_blur", in_shape=(1, 3, 256, 256), kernel_size=61, intensity=3.0, sigma_s=0.05)
    torch.manual_seed(2)
    meas = op.forward(x0.clone(), flatten=True)
    model = ocond.GuidedDenoiser(sd, cfg, op, meas, "I", x0_cov_type="convert")
    times = {}
    for tag, i in (("hi", 10), ("lo", 95)):
        s = float(sig[i])
        x = x0 + s * torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(3))
        model(x, torch.tensor([s]))                       # warm-up (thread pool, allocator)
        t0 = time.perf_counter()
        osamp.sample_heun(model, x, sig[i:i + 4])         # 3 Heun steps = 6 guided calls
        times[tag] = (time.perf_counter() - t0) / 6
    n_lo = 42
    sec_per_image = (CALLS_PER_IMAGE - n_lo) * times["hi"] + n_lo * times["lo"]
    # cross-check that the port is a fair stand-in for the real reference (within 10 %): both timed on the SAME cores in
    # the build container, where the reference can be imported (oracle/measure_ref_vs_oracle.py -> profiles/r04/ref_vs_oracle_cpu.json)
    try:
        rv = json.load(open(REF_VS_ORACLE))
        cross = {"source": "profiles/r04/ref_vs_oracle_cpu.json (python -m oracle.measure_ref_vs_oracle in the build container: the imported "
                           "reference and the oracle on the same %d cores, full-size batch-1 guided calls, best of 3)" % rv["cores"],
                 "cores": rv["cores"], **{k: {kk: v[kk] for kk in ("reference_s_per_call", "oracle_s_per_call", "oracle_over_reference")}
                                          for k, v in rv["calls"].items()}}
    except Exception as e:
        cross = {"error": repr(e)[:200]}
    return {"value": 1.0 / sec_per_image, "unit": "images/s", "cores": ncores, "kind": "port",
            "s_per_call_closed_form": round(times["hi"], 4), "s_per_call_cg": round(times["lo"], 4),
            "reference_cross_check": cross,
            "sample": "oracle (torch-CPU fp32 restatement, validated against the reference) at batch 1: 3 Heun steps = 6 "
                      "Type-I/Convert guided calls from sigma=%.3g (closed form) + 3 Heun steps = 6 calls from sigma=%.3g (CG "
                      "branch), extrapolated to 157 + 42 calls = 100 Heun steps" % (float(sig[10]), float(sig[95]))}


class PowerSampler:
    """rocm-smi package power / shader clock beside the timed region (rank 0): the workload runs the chip at its power limit, so
    the clock the MFMA peak is quoted at (2.4 GHz) is not the clock the kernels get.  Best effort: absent tool or
    unparsable output -> no `power` object."""

    def __init__(self, period=0.4):
        import threading
        self.period, self.samples, self._stop = period, [], threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        import subprocess
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10)
        card = next(iter(json.loads(r.stdout[r.stdout.index("{"):]).values()))
        pw = next((float(v) for k, v in card.items() if "Power (W)" in k), None)
        sc = next((v for k, v in card.items() if k.startswith("sclk clock speed")), None)
        mhz = float(sc.strip("()").lower().replace("mhz", "")) if sc else None
        return pw, mhz

    def _run(self):
        while not self._stop.is_set():
            try:
                self.samples.append(self._read())
            except Exception:
                pass
            self._stop.wait(self.period)

    def __enter__(self):
        self._th.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._th.join(timeout=15)

    def summary(self):
        import subprocess
        pw = sorted(p for p, _ in self.samples if p)
        ck = sorted(c for _, c in self.samples if c)
        if len(pw) < 3:
            return None
        pw, ck = pw[len(pw) // 4:], ck[len(ck) // 4:] if len(ck) > 3 else ck      # (drop the ramp-up quarter: lowest samples)
        out = {"package_power_w_median": pw[len(pw) // 2], "package_power_w_max": pw[-1], "samples": len(self.samples),
               "source": "rocm-smi --showpower --showclocks polled beside the timed region"}
        if ck:
            out["shader_clock_mhz_median"] = sorted(ck)[len(ck) // 2]
        try:
            r = subprocess.run(["rocm-smi", "--showmaxpower", "--json"], capture_output=True, text=True, timeout=10)
            card = next(iter(json.loads(r.stdout[r.stdout.index("{"):]).values()))
            out["package_power_cap_w"] = next(float(v) for k, v in card.items() if "Power" in k)
        except Exception:
            pass
        return out


def self_launch(n):
    """Re-run this command line as n ranks: python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1
    --master-port <free port> bench.py <same arguments>.  stdout / stderr are inherited, the exit code is the launcher's."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg1", help="which BASELINE.json config to run on this GPU (default cfg1 = the one the metric is quoted on; "
                                                                                   "the others: same JSON line, no extra legs, no CPU baseline)")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default: the workload's per-GPU batch, 16; cfg0: 1)")
    ap.add_argument("--no-large-batch", action="store_true", help="skip the extra throughput leg at batch 128 (N = 1 only)")
    ap.add_argument("--streams", type=int, default=2, help="part-batches per GPU, each on its own HIP stream + host thread")
    ap.add_argument("--stagger-ms", type=float, default=-1.0,
                    help="start part-batch k of a GPU k x this many milliseconds after part 0 (phase offset between the streams; INSIDE the timed region).  -1 (default) = a quarter of a "
                         "guided call's wall time (forward-only guidance: half), estimated from the last warm-up step: the UNet runs chip-filling convs at both ends of a pass and latency-bound "
                         "small-map launches in the middle, and two streams started together stay in the same phase; 0 = start together")
    ap.add_argument("--stream-prio", action="store_true", help="give every second part-batch stream the higher HIP stream priority (scheduling experiment)")
    ap.add_argument("--cu-split", action="store_true", help="give each part-batch stream its own share of the compute units (CU-masked HIP streams)")
    ap.add_argument("--dtype", choices=("bf16", "f32", "bf16x3", "f16x3"), default="f16x3",
                    help="UNet arithmetic: f16x3 (default) = fp32 storage + split-precision convs with an fp16 head (3 fp16 MFMAs per product, 22-bit operands; every call polls a two-sided "
                         "fp16-window watch and is redone in bf16x3 when it fires); bf16x3 = the same with a bf16 head (fp32 exponent range in every product): both meet north_star's 1e-3 dB against the "
                         "reference's fp32 arithmetic; f32 = exact-f32 MFMA; bf16 = the throughput mode (narrower than the reference: reported beside the headline, never as it)")
    ap.add_argument("--no-graph-leg", action="store_true", help="skip the extra leg that replays the guided calls from hipGraphs (N = 1 only)")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the extra legs that time the same workload in the other two arithmetic modes (N = 1 only)")
    ap.add_argument("--topology-only", action="store_true", help="start the ranks, report {ranks, distinct_devices, backend, devices} as one JSON line and exit "
                                                                  "(no GPU work: also runs on a CPU-only host, where the ranks rendezvous over gloo)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()
    WL = WORKLOADS[args.workload]
    if args.batch is None:
        args.batch = WL["batch"]
    if args.workload != "cfg1":      # the extra legs and the CPU baseline belong to the headline workload
        args.no_large_batch = args.no_graph_leg = args.no_f32_leg = args.no_cpu_baseline = True

    # `python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start the N ranks ourselves, one process per
    # GPU under torch.distributed.run on the loopback address, pass the arguments through and relay rank 0's JSON line.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    import kdip_amd._lib as L
    import kdip_amd.unet as ku
    import kdip_amd.condition as kc
    import kdip_amd.measurements as km
    import kdip_amd.sampling as ks
    from kdip_amd.evaluation import DistEnv

    env = DistEnv()
    if env.world_size != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env.world_size} (run `python bench.py --gpus N` on its own, or under a launcher with matching --nproc-per-node)")
    if args.topology_only:
        topo = env.topology()
        if env.is_main_process:
            print(json.dumps({"n_gpus": args.gpus, **topo}))
        env.barrier()
        return
    L.require_gpu()
    dev = env.device
    B, S, rank = args.batch, 256, env.rank
    lib = L.load()

    D = ku.GaussianDiffusionTables()
    NS = WL["nsteps"]                                            # sampler steps of the workload's schedule (100 Heun | 20 Euler)
    heun = WL["sampler"] == "heun"
    calls_per_image = 2 * NS - 1 if heun else NS
    ARCH = ku.FFHQ_CONFIG if WL["arch"] == "FFHQ" else ku.IMAGENET_CONFIG
    sigmas = ks.get_sigmas_karras(NS, 0.01, 80, rho=7.0, device=dev)
    sig = sigmas.detach().cpu()
    sd = ku.synthetic_state_dict(seed=0, out_cov=bool(WL.get("ortho")), **ARCH)      # random-init weights of the named architecture (no checkpoint offline)
    if os.environ.get("KDIP_EXP_ZERO_WEIGHTS"):                  # diagnostic only: identical instruction stream on all-zero conv weights
        sd = {k: (v * 0 if k.endswith("weight") and v.dim() == 4 else v) for k, v in sd.items()}

    # ---- the per-GPU batch is split into `--streams` part-batches, each with its own UNet handle (weights + workspace),
    # operator context, HIP stream and host thread: images are independent problems, so while one part is in its
    # HBM-bound GroupNorm passes or waits for a CG convergence flag, the other part's convs have the MFMA pipes.
    def build_parts(Btot, nstreams):
        nstreams = max(1, min(nstreams, Btot))
        parts = []
        for k in range(nstreams):
            Bk = Btot // nstreams + (1 if k < Btot % nstreams else 0)
            den, op, x0, meas = build_problem(WL, args.dtype, dev, Bk, sd, D, seed=1000 * rank + 100 * k)
            noise = torch.randn(Bk, 3, S, S, device=dev, generator=torch.Generator(device=dev).manual_seed(3 + 1000 * rank + 100 * k))
            if nstreams > 1 and args.cu_split:      # stream k owns CU indices [32 k / n, 32 (k + 1) / n) of every XCD (8 mask bits per CU index)
                lo, hi = 32 * k // nstreams, 32 * (k + 1) // nstreams
                bits = sum(0xff << (8 * j) for j in range(lo, hi))
                words = (C.c_uint * 8)(*[(bits >> (32 * w)) & 0xffffffff for w in range(8)])
                hs = C.c_void_p()
                L.check(lib.kdip_stream_create_cu_mask(dev.index if hasattr(dev, "index") and dev.index is not None else torch.cuda.current_device(), words, 8, C.byref(hs)))
                stream = torch.cuda.ExternalStream(hs.value, device=dev)
            else:
                # --stream-prio: odd part-batch streams get the higher HIP stream priority (experiment: does an asymmetric pair overlap better?)
                stream = (torch.cuda.Stream(device=dev, priority=(-1 if (args.stream_prio and k % 2) else 0)) if nstreams > 1 else torch.cuda.current_stream())
            parts.append(dict(den=den, x0=x0, noise=noise, stream=stream, B=Bk))
        torch.cuda.synchronize()
        return parts

    def start_state(pt, i):
        return (pt["x0"] + float(sig[i]) * pt["noise"]).contiguous() if i > 0 else (pt["noise"] * float(sig[0])).contiguous()

    def run_part(pt, steps, chain):
        x = start_state(pt, 0)
        for i in steps:
            if not chain or i == 0:
                x = start_state(pt, i)
            x = ks.heun_step(pt["den"], x, sig, i) if heun else ks.sample_euler(pt["den"], x, sig[i:i + 2], disable=True)
        return x

    stagger = {"ms": max(args.stagger_ms, 0.0)}            # (auto: set by timed_run from its last warm-up step)

    def run_all(parts, steps, chain):
        from kdip_amd.evaluation import run_on_streams
        outs = run_on_streams([lambda pt=pt: run_part(pt, steps, chain) for pt in parts], [pt["stream"] for pt in parts], dev,
                              delays=[k * stagger["ms"] * 1e-3 for k in range(len(parts))] if stagger["ms"] > 0 else None)
        torch.cuda.synchronize()
        return torch.cat(outs)

    def timed_run(parts, steps, chain, warmup):
        """warm-up (one closed-form and one CG step per pair: allocates the workspaces), then barrier-bracketed timing incl. the
        one collective of the path (RCCL all_gather); returns max-over-ranks seconds."""
        if warmup > 0:
            wsteps = [NS // 10 if w % 2 == 0 else NS * 95 // 100 for w in range(warmup)]
            if len(wsteps) > 1:
                run_all(parts, wsteps[:-1], False)
            torch.cuda.synchronize()
            tw = time.perf_counter()
            run_all(parts, wsteps[-1:], False)
            if args.stagger_ms < 0 and len(parts) > 1 and len(wsteps) > 1:      # phase offset = a quarter (forward-only guidance: half) of one guided call with all streams running
                call_ms = (time.perf_counter() - tw) * 1e3 / (2 if heun else 1)
                stagger["ms"] = round(call_ms / (2 if WL["guidance"] == "II" else 4), 2)
        torch.cuda.synchronize()
        env.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = run_all(parts, steps, chain)                       # (ends with a device synchronisation)
        t1 = time.perf_counter()
        hat = env.gather(x)                                    # the path's one collective: all_gather of the results (RCCL over xGMI)
        torch.cuda.synchronize()
        timed_run.gather_ms = (time.perf_counter() - t1) * 1e3
        env.barrier()
        elapsed = env.max_over_ranks(time.perf_counter() - t0)
        assert torch.isfinite(hat).all() and hat.shape[0] == env.world_size * x.shape[0]
        return elapsed

    full_run = args.steps % NS == 0 and args.steps > 0
    idx = step_indices(args.steps, NS, last_single=heun)
    parts = build_parts(B, args.streams)
    S_ = len(parts)
    sampler = PowerSampler() if env.is_main_process and not os.environ.get("KDIP_NO_POWER") else None
    if sampler:
        with sampler:
            elapsed = timed_run(parts, idx, full_run, args.warmup)
    else:
        elapsed = timed_run(parts, idx, full_run, args.warmup)
    den, x0 = parts[0]["den"], parts[0]["x0"]             # the roofline leg profiles part 0 alone

    gather_ms = env.max_over_ranks(getattr(timed_run, "gather_ms", 0.0))
    topo = env.topology()                                      # (after the timed region)
    ms_per_step = elapsed / args.steps * 1e3
    images_per_s = env.world_size * B / (ms_per_step * NS / 1e3)
    out = {
        "metric": "images/sec (256x256 FFHQ Gaussian deblur, Type-I + Convert, 100 Heun steps)" if args.workload == "cfg1" else f"images/sec ({WL['label'].split(':')[0]})",
        "value": round(images_per_s, 5), "unit": "images/s", "n_gpus": env.world_size, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": f"synthetic (seeded smooth images, random-init {WL['arch']}-architecture weights)",
        "config": {"workload": WL["label"] + ", batch " + str(B) + " per GPU", "reference": WL["ref"],
                   "global_batch": env.world_size * B, "per_gpu_batch": B, "streams_per_gpu": S_, "stream_phase_offset_ms": stagger["ms"], "images_per_launch": parts[0]["B"],
                   "calls_per_image": calls_per_image,
                   "timed_steps": f"full {NS}-step sampler run" if full_run else (f"two-call Heun steps spread evenly over steps 0..{NS - 2} of the {NS}-step schedule (the single-call final step is never in a subset)" if heun else f"Euler steps spread evenly over the {NS}-step schedule"),
                   "parallelism": f"dp{env.world_size} (independent images, one all_gather at the end)"},
        "achieved_tflops_whole_step": round((2 if heun else 1) * B * WL["gflop_call"] / ms_per_step, 2),
        # multi-GPU evidence (k_diffusion/evaluation.py:53-63): N ranks on N distinct devices, the collective's backend, and the
        # time of the final all_gather (inside the timed region; max over ranks)
        "ranks": topo["ranks"], "distinct_devices": topo["distinct_devices"], "backend": topo["backend"], "devices": topo["devices"],
        "gather_ms": round(gather_ms, 3), "gather_bytes_per_rank": B * 3 * S * S * 4,
    }
    try:      # peak device workspace of the UNet handles (persist + scratch + statistics arenas planned for this batch) and what torch holds beside them
        out["workspace_gb"] = {"unet_handles": round(sum(unet_of(pt["den"]).workspace_bytes(pt["B"]) for pt in parts) / 1e9, 3),
                               "torch_peak_allocated": round(torch.cuda.max_memory_allocated() / 1e9, 3)}
    except Exception as e:
        out["workspace_gb"] = {"error": repr(e)[:120]}
    if args.dtype == "f16x3":       # fp16-headed split: calls of the timed region that had to be redone bf16-headed because an operand left the fp16 window (UNetModel.guarded)
        out["x3_fallbacks"] = sum(unet_of(pt["den"]).x3_fallbacks for pt in parts)
        out["x3_degraded_fallbacks"] = sum(unet_of(pt["den"]).x3_degraded for pt in parts)
    if args.dtype == "bf16x3":      # every conv product of the timed region carried its full split precision (no operand left the fp16 window of the tail planes)
        sat = 0
        for pt in parts:
            with torch.cuda.stream(pt["stream"]):
                sat |= unet_of(pt["den"]).x3_saturated()
        out["x3_saturated"] = sat       # 0 = every product of every conv kept its full split precision (kdip_unet_x3_saturated)
    pw = sampler.summary() if sampler else None
    if pw:
        out["power"] = pw

    # ---- roofline leg: per-launch HIP-event timing of the conv kernels on two representative steps
    if not args.no_roofline and env.is_main_process:
        L.check(lib.kdip_profile_enable(1))
        for i in (NS // 10, NS * 95 // 100):
            run_part(parts[0], [i], False)
        # ... and one pass over the operator / transform kernels of the path that this workload (Gaussian deblur, pixel basis) does
        # not touch, on the same 8 x 3 x 256 x 256 planes: motion blur (FFT model), 4x SR (Resizer + its adjoint + FFT solver model),
        # inpainting (gather / scatter / mask), Haar DWT / IDWT -- north_star judges them by HBM GB/s (`hbm_bound_classes.op_*`)
        try:
            from kdip_amd.transforms import OrthoTransform
            xo = x0.contiguous()
            np.random.seed(0)
            ops = {n: km.get_operator(n, device=dev, **kw) for n, kw in (
                ("motion_blur", dict(in_shape=(1, 3, S, S), kernel_size=61, intensity=0.5, sigma_s=0.05)),
                ("super_resolution", dict(in_shape=(1, 3, S, S), scale_factor=4, sigma_s=0.05)),
                ("inpainting", dict(sigma_s=0.05, mask_opt=dict(mask_type="random", mask_prob_range=(0.5, 0.5), image_size=S))))}
            dwt = OrthoTransform("dwt")
            for _ in range(3):
                ym = ops["motion_blur"].forward(xo, noiseless=True); ops["motion_blur"].transpose(ym)
                ys = ops["super_resolution"].forward(xo, noiseless=True); ops["super_resolution"].forward_adjoint(ys); ops["super_resolution"].transpose(ys)
                yi, yif = ops["inpainting"].forward(xo, flatten=True); ops["inpainting"].transpose(yif, flatten=True)
                dwt.inv(dwt(xo))
        except Exception as e:
            out["roofline_operator_pass_error"] = repr(e)[:200]
        torch.cuda.synchronize()
        n = lib.kdip_profile_num_classes()
        ms = (C.c_double * n)(); fl = (C.c_double * n)(); by = (C.c_double * n)(); la = (C.c_long * n)()
        L.check(lib.kdip_profile_report(ms, fl, by, la))
        base = [(ms[j], by[j], la[j]) for j in range(n)]
        # ... the same operator kernels once more on 64 images (192 planes, 50 MB per tensor): at 8 images a pass moves 6 MB and is
        # launch-bound, this is the bandwidth the kernels reach when the launch is amortised (reported as `op_bandwidth_at_64_images`;
        # every other number of `roofline` comes from the report taken above, before this pass)
        op_big = None
        try:
            xb = smooth_image(64, S, seed=7).to(dev)
            gb = km.get_operator("gaussian_blur", device=dev, in_shape=(1, 3, S, S), kernel_size=61, intensity=3.0, sigma_s=0.05)
            for _ in range(3):
                yg = gb.forward(xb, noiseless=True); gb.transpose(yg)
                ym = ops["motion_blur"].forward(xb, noiseless=True); ops["motion_blur"].transpose(ym)
                ys = ops["super_resolution"].forward(xb, noiseless=True); ops["super_resolution"].transpose(ys)
                yi, yif = ops["inpainting"].forward(xb, flatten=True); ops["inpainting"].transpose(yif, flatten=True)
                dwt.inv(dwt(xb))
            torch.cuda.synchronize()
            ms2 = (C.c_double * n)(); fl2 = (C.c_double * n)(); by2 = (C.c_double * n)(); la2 = (C.c_long * n)()
            L.check(lib.kdip_profile_report(ms2, fl2, by2, la2))
            op_big = {lib.kdip_profile_class_name(j).decode(): {"GBps": round((by2[j] - base[j][1]) / max(ms2[j] - base[j][0], 1e-9) / 1e6, 1),
                                                                "launches": int(la2[j] - base[j][2]), "avg_launch_us": round((ms2[j] - base[j][0]) * 1e3 / max(la2[j] - base[j][2], 1), 1)}
                      for j in range(n) if la2[j] - base[j][2] > 0 and lib.kdip_profile_class_name(j).decode().startswith("op_")}
            del xb
        except Exception as e:
            op_big = {"error": repr(e)[:200]}
        # dominant kernel = the (kernel class, layer shape) with the largest total time in the profiled pass
        import csv, tempfile, collections
        dump = os.environ.get("KDIP_PROFILE_DUMP")
        if dump:                            # one file per arithmetic mode: the sub-legs below run this script again with the same environment
            root_, ext_ = os.path.splitext(dump)
            dump = f"{root_}.{args.dtype}{ext_ or '.csv'}"
        else:
            dump = os.path.join(tempfile.gettempdir(), f"kdip_conv_dump_{os.getpid()}.csv")
        L.check(lib.kdip_profile_dump(dump.encode()))
        grp = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
        for r in csv.DictReader(open(dump)):
            if not r["class"].startswith("conv"):
                continue
            key = (r["class"], r["tag"], r["d0"], r["d1"], r["d2"], r["d3"])
            grp[key][0] += 1; grp[key][1] += float(r["us"]); grp[key][2] += float(r["gflop"]); grp[key][3] += float(r["mbytes"])
        key, (cnt, us, gf, mb) = max(grp.items(), key=lambda kv: kv[1][1])
        tflops = gf / us * 1e3 if us > 0 else 0.0          # GFLOP / us = PFLOP/s
        # HBM traffic of this (kernel, fusion mode, layer shape): rocprofv3 --pmc passes over the same in-network launches
        # (tools/pmc_innetwork.sh -> profiles/r04_pmc_innetwork.json; FETCH_SIZE x 2 + WRITE_SIZE per MI355X_MICROARCH.md)
        traffic, traffic_source, pmc_extra = None, None, {}
        cands = ["r06_pmc_innetwork_" + args.dtype, "r05_pmc_innetwork_" + args.dtype] + (["r06_pmc_innetwork", "r05_pmc_innetwork", "r04_pmc_innetwork", "r03_pmc_innetwork", "r02_pmc_innetwork"] if args.dtype == "bf16" else [])
        pmc = next((f for f in (os.path.join(ROOT, "profiles", t + ".json") for t in cands) if os.path.exists(f)), "")
        if pmc:
            try:
                shapes_, base_ = json.load(open(pmc))["shapes"], "|".join(key[1:])
                # (first-generation kernel: the PMC file keys one entry per template instantiation, "<tag|shape>|<template arguments>")
                cand_ = [v for k_, v in shapes_.items() if k_ == base_ or k_.startswith(base_ + "|")]
                e = max(cand_, key=lambda v: v.get("launches_per_pass", 0) * v.get("mean_launch_us_under_pmc", 0)) if cand_ else None
                if e:
                    traffic = e.get("hbm_bytes_per_launch")
                    traffic_source = ("profiles/" + os.path.basename(pmc) + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the in-network launches "
                                      "of this kernel + fusion mode + layer shape (tools/pmc_innetwork.sh on one guided Heun step of this workload); "
                                      "not re-measured in this run.  bytes = FETCH_SIZE x 2 + WRITE_SIZE x 1: factors CALIBRATED on known byte counts in this "
                                      "kernel's own access patterns (profiles/r06/fetch_calibration.txt: 2.000 / 1.996 / 2.000 for reads, 1.000 for writes)")
                    pmc_extra = {k: e[k] for k in ("traffic_over_algorithmic", "mfma_busy_frac", "shader_clock_ghz", "lds_bank_conflict_frac_of_lds_active",
                                                   "l2_hit_rate", "FETCH_SIZE_KiB_raw", "WRITE_SIZE_KiB_raw", "algorithmic_bytes") if k in e}
                    pmc_extra["fetch_size_factor"] = 2.0; pmc_extra["write_size_factor"] = 1.0
            except Exception:
                traffic = None
        fusion = {"conv3": "plain", "conv3_gnf": "GroupNorm+SiLU fused into the input staging", "conv3_gnb": "GroupNorm backward fused into the input staging"}
        tagparts = key[1].split("_")
        base = "_".join(tagparts[:2]) if len(tagparts) > 1 and tagparts[1] in ("gnf", "gnb") else tagparts[0]
        desc = fusion.get(base, "first-generation kernel (conv.hip)" + (": split precision, 1 v_mfma_f32_32x32x16_bf16 + 2 v_mfma_f32_32x32x16_f16 per product" if args.dtype == "bf16x3" else (": fp16-headed split precision, 3 v_mfma_f32_32x32x16_f16 per product, 1 x 4 waves with a row-reuse K loop on the chip-filling launches (every call polled, redone bf16-headed outside the fp16 window)" if args.dtype == "f16x3" else ""))) + ("; GroupNorm forward sums of the output in the epilogue" if "s1" in tagparts else "") + \
            ("; GroupNorm backward sums of the output in the epilogue" if "s2" in tagparts else "") + ("; residual add" if "res" in tagparts else "")
        names = [lib.kdip_profile_class_name(j).decode() for j in range(n)]
        k = max((j for j in range(n) if names[j].startswith("conv")), key=lambda j: ms[j])
        out["roofline"] = {
            "kernel": f"{key[1]} ({'conv3_kernel, csrc/conv3.hip' if key[1].startswith('conv3') else 'conv_igemm_kernel, csrc/conv.hip'}, {'v_mfma_f32_32x32x16_f16' if args.dtype == 'f16x3' else 'v_mfma_f32_32x32x16_bf16'}: {desc}), "
                      f"layer B={key[2]} {key[4]}->{key[5]} ch @ {key[3]}x{key[3]}",
            "bound": "mfma", "achieved": round(tflops, 2), "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tflops / BF16_MFMA_PEAK_TFLOPS, 4),
            # `achieved` counts ALGORITHMIC flops (one multiply-add per MAC).  The split-precision mode issues three 16-bit MFMAs
            # per product (the exact-f32 mode: the 8x slower fp32 MFMA), so the matrix pipes are busier than `frac` says by this factor:
            "mfma_instructions_per_product": {"bf16x3": 3, "f16x3": 3, "bf16": 1, "f32": 1}[args.dtype],
            "mfma_work_frac_of_peak": round({"bf16x3": 3.0, "f16x3": 3.0, "bf16": 1.0, "f32": 1.0}[args.dtype] * tflops / BF16_MFMA_PEAK_TFLOPS, 4) if args.dtype != "f32" else None, "traffic": traffic, "traffic_source": traffic_source, "pmc_in_network": pmc_extra,
            "launches": cnt, "avg_launch_us": round(us / cnt, 2), "algorithmic_gflop_per_launch": round(gf / cnt, 3),
            "algorithmic_bytes_per_launch": round(mb / cnt * 1e6),
            "share_of_profiled_conv_time": round(us / max(sum(v[1] for v in grp.values()), 1e-9), 3),
            # `peak` is the nominal dense bf16 figure (2.4 GHz); the chip sustains 1.8 - 2.1 GHz under these launches (GRBM_GUI_ACTIVE /
            # duration in pmc_in_network; s_memtime / wall clock in tools/conv3_phases.py: 1.78 GHz), so the same launch also as a
            # fraction of the MFMA rate at the clock it actually ran at
            "frac_of_peak_at_measured_clock": (round(tflops / (BF16_MFMA_PEAK_TFLOPS * pmc_extra["shader_clock_ghz"] / 2.4), 4)
                                               if pmc_extra.get("shader_clock_ghz") else None),
            "top_conv_launch_groups": [{"tag": kk[1], "B": int(kk[2]), "HW": int(kk[3]), "Cin": int(kk[4]), "Cout": int(kk[5]), "launches": vv[0],
                                        "avg_launch_us": round(vv[1] / vv[0], 1), "tflops": round(vv[2] / vv[1] * 1e3, 1)}
                                       for kk, vv in sorted(grp.items(), key=lambda kv: -kv[1][1])[:8]],
            "class_aggregate": {"kernel_class": lib.kdip_profile_class_name(k).decode(), "tflops": round(fl[k] / max(ms[k], 1e-9) / 1e9, 2),
                                "launches": int(la[k]), "avg_launch_us": round(ms[k] * 1e3 / max(la[k], 1), 2)},
            "all_conv_classes": {lib.kdip_profile_class_name(j).decode(): {"ms": round(ms[j], 3), "tflops": round(fl[j] / max(ms[j], 1e-9) / 1e9, 2), "launches": int(la[j])}
                                 for j in range(n) if la[j] > 0 and names[j].startswith("conv")},
            # the GroupNorm streaming passes and the operator / transform / point-wise kernels (op_*, pointwise): algorithmic bytes /
            # HIP-event time against HBM (~8 TB/s peak, ~6.3 TB/s achievable); the op_* launches move 6 - 25 MB each, i.e. they are
            # launch- / latency-bound at this batch
            "hbm_peak_GBps": 8000.0, "op_bandwidth_at_64_images": op_big,
            "hbm_bound_classes": {names[j]: {"ms": round(ms[j], 3), "GBps": round(by[j] / max(ms[j], 1e-9) / 1e6, 1), "launches": int(la[j])}
                                  for j in range(n) if la[j] > 0 and not names[j].startswith("conv")},
        }
        L.check(lib.kdip_profile_enable(0))

    # ---- extra leg (N = 1): the same timed steps with every closed-form guided call replayed from a hipGraph captured once per
    # (sigma, part-batch) -- what a server that runs this schedule for batch after batch does (kdip_amd/graphs.py); the CG-branch
    # calls stay eager.  Captures happen in an untimed pass, sequentially per part.
    if not args.no_graph_leg and env.world_size == 1 and args.dtype != "f32":
        try:
            from kdip_amd.graphs import GraphedDenoiser
            for pt in parts:
                pt["eager_den"], pt["den"] = pt["den"], GraphedDenoiser(pt["den"])
                with torch.cuda.stream(pt["stream"]):
                    run_part(pt, sorted(set(idx)), False)                      # capture pass
                torch.cuda.synchronize()
            el = timed_run(parts, idx, full_run, 0)
            rep = sum(pt["den"].replays for pt in parts)
            eag = sum(pt["den"].eager_calls for pt in parts)
            unconv = sum(pt["den"].cg_unconverged() for pt in parts)     # fixed-trip CG replays whose last residual check still found an active sample
            out["hipgraph_replay"] = {"ms_per_step": round(el / args.steps * 1e3, 3), "value": round(env.world_size * B / (el / args.steps * 100), 5), "unit": "images/s",
                                      "graphs": sum(len(pt["den"]._graphs) for pt in parts), "replayed_calls_incl_capture_pass": rep, "eager_calls": eag,
                                      "cg_fixed_trip_graphs": sum(len(pt["den"].cg_trips) for pt in parts), "cg_unconverged_replays": unconv,
                                      "note": "same steps and protocol as `value`; every guided call replayed from a hipGraph captured in an untimed pass -- "
                                              "closed-form calls as they are, CG-branch calls as fixed-trip solves (1.5 x the warm-up call's iterations + 4, no host read)"}
            for pt in parts:
                pt["den"] = pt["eager_den"]
        except Exception as e:
            out["hipgraph_replay"] = {"error": repr(e)[:300]}

    # ---- extra leg (N = 1, default workload only): the same sampler run at the throughput-optimal batch.  BASELINE configs[1]
    # fixes the batch at 16 images per GPU (that is `value`); per-image cost keeps falling until ~128 images per GPU.  Run as a
    # fresh process (this one is idle meanwhile): re-using this process after its own workspaces were freed measured 6 % slower.
    if not args.no_large_batch and env.world_size == 1 and full_run and B == 16:
        import subprocess
        del parts, den, x0
        torch.cuda.empty_cache()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--dtype", args.dtype, "--batch", "128", "--streams", str(args.streams), "--steps", str(args.steps),
                                "--warmup", "2", "--no-cpu-baseline", "--no-roofline", "--no-large-batch", "--no-f32-leg", "--no-graph-leg"], capture_output=True, text=True, timeout=1500)
            big = json.loads(r.stdout.strip().splitlines()[-1])
            out["throughput_at_batch_128"] = {"value": big["value"], "unit": "images/s", "per_gpu_batch": 128,
                                              "streams_per_gpu": big["config"]["streams_per_gpu"], "ms_per_step": big["ms_per_step"],
                                              "note": "same code path and timing protocol (python bench.py --batch 128), 128 images per GPU instead of the 16 of BASELINE configs[1]"}
        except Exception as e:            # the headline measurement above must survive a failure of the optional leg
            out["throughput_at_batch_128"] = {"error": repr(e)[:200]}

    # ---- extra legs (N = 1): the SAME workload, protocol and code path in the other two arithmetic modes, each as a fresh process.
    # `value` is the bf16x3 mode (fp32 storage, split-precision convs: the fast mode that meets the 1e-3 dB tolerance against the
    # reference's fp32 arithmetic, tests/test_parity_gpu.py / test_x3_gpu.py / test_fullsize_gpu.py); bf16 is the throughput mode --
    # narrower than the reference (utils_model.py:364 use_fp16=False), reported beside the headline, never as it; f32 = exact-f32 MFMA.
    if not args.no_f32_leg and env.world_size == 1 and B == 16:
        import subprocess

        def sub_leg(dtype, steps, flags):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--dtype", dtype, "--batch", str(B), "--streams", str(args.streams), "--steps", str(steps),
                                "--warmup", "2", "--no-cpu-baseline", "--no-large-batch", "--no-f32-leg"] + flags, capture_output=True, text=True, timeout=900)
            return json.loads(r.stdout.strip().splitlines()[-1])

        def leg_roofline(rf, per_product):
            if not rf:
                return None
            return {"kernel": rf.get("kernel"), "achieved": rf.get("achieved"), "unit": "TFLOP/s", "frac": rf.get("frac"), "avg_launch_us": rf.get("avg_launch_us"),
                    "mfma_work_frac_of_peak": round(per_product * rf["achieved"] / BF16_MFMA_PEAK_TFLOPS, 4) if rf.get("achieved") else None,
                    "all_conv_classes": rf.get("all_conv_classes"),
                    "hbm_bound_classes": {k: v for k, v in (rf.get("hbm_bound_classes") or {}).items() if k.startswith("gn")}}
        others = [d for d in ("f16x3", "bf16x3", "bf16", "f32") if d != args.dtype]
        for d in others:
            name = {"f16x3": "f16x3_parity_mode", "bf16x3": "bf16x3_parity_mode", "bf16": "bf16_throughput_mode", "f32": "f32_parity_mode"}[d]
            try:
                if d == "f32":
                    leg = sub_leg("f32", 4, ["--no-roofline", "--no-graph-leg"])
                else:
                    leg = sub_leg(d, args.steps if full_run else 20, [] if d == "bf16" else ["--no-graph-leg"])
                o = {"value": leg["value"], "unit": "images/s", "dtype": d, "ms_per_step": leg["ms_per_step"], "steps": leg["steps"],
                     "over_headline": round(leg["value"] / images_per_s, 3), "achieved_tflops_whole_step": leg["achieved_tflops_whole_step"]}
                if d != "f32":
                    o["roofline"] = leg_roofline(leg.get("roofline"), 3.0 if d in ("bf16x3", "f16x3") else 1.0)
                if d == "bf16":
                    o["hipgraph_replay"] = leg.get("hipgraph_replay")
                    o["note"] = ("same workload, protocol and code path (python bench.py --dtype bf16): bf16 activations and MFMA inputs -- NOT tolerance-compliant (|dPSNR| ~1e-2 dB "
                                 "against the f32 arithmetic end to end, per-call PSNR floors in tests/test_fullsize_gpu.py::test_e2e_teacher_forced); a throughput figure, not the headline")
                elif d == "f16x3":
                    o["x3_fallbacks"] = leg.get("x3_fallbacks")
                    o["note"] = ("same workload, protocol and code path (python bench.py --dtype f16x3): fp32 activations, every conv as 3 fp16 MFMAs per product (fp16 head + fp16 tails, 11 + 11-bit "
                                 "operands), fp32 accumulation; every guided call polls the fp16-window flag and is redone in the bf16x3 arithmetic when an operand left the window (x3_fallbacks)")
                elif d == "f32":
                    o["note"] = "same workload, protocol and code path (python bench.py --dtype f32 --steps 4): fp32 activations + v_mfma_f32_32x32x2_f32, the reference's own arithmetic"
                else:
                    o["note"] = "same workload, protocol and code path (python bench.py --dtype bf16x3): fp32 activations, every conv as 1 bf16 + 2 fp16 MFMAs per product, fp32 accumulation"
                out[name] = o
            except Exception as e:
                out[name] = {"error": repr(e)[:200]}

    # ---- CPU baseline (rank 0, single-GPU runs only; bounded sample)
    if not args.no_cpu_baseline and env.is_main_process and env.world_size == 1:
        cb = cpu_baseline(sig)
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_baseline"] = round(images_per_s / cb["value"], 1)

    if env.is_main_process:
        print(json.dumps(out))
    env.barrier()


if __name__ == "__main__":
    main()
