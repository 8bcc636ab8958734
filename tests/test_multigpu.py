"""Two-GPU check of the N > 1 path on real hardware (marker `gpu2`: skipped unless the node shows two MI355X).  One process per
GPU over RCCL (torch.distributed backend "nccl"), contiguous shards, no communication until the final all_gather
(k_diffusion/evaluation.py:53-63): compute_features over 2 ranks must equal the single-process result, the ranks must sit on
distinct devices, and `bench.py --gpus 2` must report ranks = distinct_devices = 2 with backend nccl."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from kdip_amd.evaluation import DistEnv, compute_features, shard_range
import kdip_amd.unet as ku, kdip_amd.external as ke
env = DistEnv()                                   # nccl (= RCCL) on GPUs
assert env.world_size == 2 and env.device.type == "cuda"
topo = env.topology()
assert topo["ranks"] == 2 and topo["distinct_devices"] == 2 and topo["backend"] == "nccl", topo
# every rank denoises its contiguous shard of the same 6 seeded inputs with the same tiny UNet; the gathered result must equal
# the whole batch denoised by one process (rank 0 recomputes it)
from oracle import unet as ounet
cfg = ounet.UNetConfig(**ounet.TINY)
sd = ounet.init_state_dict(cfg, seed=0)
m = ku.UNetModel(image_size=64, model_channels=32, num_res_blocks=1, attention_resolutions="32", channel_mult=(1, 2), dtype="f32", device=env.device)
m.load_state_dict(sd)
den = ke.OpenAIDenoiser(m, ku.GaussianDiffusionTables())
xs = torch.randn(6, 3, 64, 64, generator=torch.Generator().manual_seed(0))
lo, hi = shard_range(6, env.rank, env.world_size)
sig = torch.full((3,), 1.5, device=env.device)
def sample_fn(n):
    return den(xs[lo:lo + n].to(env.device), sig[:n])
out = compute_features(env, sample_fn, lambda x: x, 6, 3)
assert out.shape == (6, 3, 64, 64)
if env.is_main_process:
    ref = torch.cat([den(xs[i:i + 3].to(env.device), sig) for i in (0, 3)])
    err = float((out - ref).abs().max())
    assert err < 1e-4, err                        # fp64-atomic order noise only
env.barrier()
print("rank", env.rank, "ok")
'''


@pytest.mark.gpu2
def test_compute_features_rccl_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", str(script), ROOT]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == 2


def test_bench_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the driver's command line) starts its two ranks itself; `--topology-only`
    stops after the rendezvous, so this runs on the CPU container too (gloo)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--topology-only"], capture_output=True, text=True, timeout=600,
                       cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["backend"] in ("gloo", "nccl"), line


@pytest.mark.gpu2
def test_bench_two_gpus_reports_topology():
    # no torchrun in front: bench.py launches its ranks itself (rank 0 prints the one JSON line)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1800, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["distinct_devices"] == 2 and line["backend"] == "nccl", line
    assert line["gather_ms"] > 0 and line["scaling"] == "weak"
