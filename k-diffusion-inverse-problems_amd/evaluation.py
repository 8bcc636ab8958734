"""Multi-process evaluation helper -- k_diffusion/evaluation.py:53-63 (`compute_features`).

One process per GPU; the n samples are split over the processes, sampled independently
(no communication inside the sampler loop) and gathered once at the end with a
torch.distributed all_gather (backend "nccl" = RCCL over xGMI on MI355X, "gloo" on CPU).
`DistEnv` is the minimal stand-in for the `accelerate.Accelerator` attributes the
reference uses (num_processes, is_main_process, gather, device).
"""
import math
import os

import torch
import torch.distributed as dist


class DistEnv:
    def __init__(self, backend=None, init=True):
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        use_cuda = torch.cuda.is_available()
        if use_cuda:
            torch.cuda.set_device(self.local_rank)
        self.device = torch.device("cuda", self.local_rank) if use_cuda else torch.device("cpu")
        if init and self.world_size > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            backend = backend or ("nccl" if use_cuda else "gloo")
            kw = {}
            if backend == "nccl":
                kw["device_id"] = self.device
            dist.init_process_group(backend, rank=self.rank, world_size=self.world_size, **kw)

    @property
    def num_processes(self):
        return self.world_size

    @property
    def is_main_process(self):
        return self.rank == 0

    is_local_main_process = is_main_process

    def gather(self, t):
        """all_gather along dim 0 (accelerator.gather)."""
        if self.world_size == 1:
            return t
        t = t.contiguous()
        out = [torch.empty_like(t) for _ in range(self.world_size)]
        dist.all_gather(out, t)
        return torch.cat(out)

    def barrier(self):
        if self.world_size > 1:
            dist.barrier()

    def device_identity(self):
        """What physical device this rank computes on: (hostname, PCI bus id | device UUID | index) -- distinct across the ranks
        of one node iff every rank has its own GPU."""
        import socket
        ident = f"cpu:{os.getpid()}"
        if self.device.type == "cuda":
            p = torch.cuda.get_device_properties(self.device)
            ident = None
            for attr in ("pci_bus_id", "uuid"):
                v = getattr(p, attr, None)
                if v is not None and str(v) not in ("", "0"):
                    ident = (f"pci:{getattr(p, 'pci_domain_id', 0)}:{v}:{getattr(p, 'pci_device_id', 0)}" if attr == "pci_bus_id" else f"{attr}:{v}")
                    break
            if ident is None:
                ident = f"index:{self.device.index}"
        return (socket.gethostname(), ident)

    def topology(self):
        """{"ranks", "distinct_devices", "backend", "devices"}: every rank's device identity gathered to all ranks (one small
        all_gather_object, outside any timed region) -- evidence that an N-rank run really used N devices over RCCL."""
        me = self.device_identity()
        if self.world_size == 1:
            ids = [me]
            backend = "none (single process)"
        else:
            ids = [None] * self.world_size
            dist.all_gather_object(ids, me)
            backend = dist.get_backend()
        return {"ranks": self.world_size, "distinct_devices": len(set(ids)), "backend": backend, "devices": [f"{h}/{d}" for h, d in ids]}

    def max_over_ranks(self, value):
        if self.world_size == 1:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])


def shard_range(n, rank, world):
    """Contiguous split of n items: rank r gets [lo, hi)."""
    per = math.ceil(n / world)
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


def compute_features(accelerator, sample_fn, extractor_fn, n, batch_size):
    """Each process draws ceil(n / P) samples in chunks of `batch_size`; chunks are
    all-gathered and the concatenation truncated to n (evaluation.py:53-63)."""
    n_per_proc = math.ceil(n / accelerator.num_processes)
    feats_all = []
    try:
        for i in range(0, n_per_proc, batch_size):
            cur_batch_size = min(n - i, batch_size)
            samples = sample_fn(cur_batch_size)[:cur_batch_size]
            feats_all.append(accelerator.gather(extractor_fn(samples)))
    except StopIteration:
        pass
    return torch.cat(feats_all)[:n]


def psnr(hat_x0, x0):
    """PSNR of compute_metrics (sample_condition_openai.py:41-44): images mapped to [0,1]."""
    a = (hat_x0 / 2 + 0.5).clip(0, 1)
    b = (x0 / 2 + 0.5).clip(0, 1)
    mse = ((a - b) ** 2).flatten(1).mean(dim=1)
    return 10 * torch.log10(1.0 / mse)


def run_on_streams(fns, streams=None, device=None, delays=None):
    """Run the callables `fns[k]()` concurrently, each inside its own HIP stream context on its own host thread, and return
    their results in order (exceptions are re-raised).  Independent part-batches driven this way overlap one part's
    HBM-bound passes and host-side convergence waits with another part's MFMA-bound convs (bench.py --streams,
    sample_condition.py --streams).  Every callable must use its own UNet handle and operator context: library handles
    are per (thread, stream), never shared.
    `delays[k]` (seconds, optional): callable k starts that much later -- a phase offset between the part-batches: the UNet runs its
    large maps (chip-filling convs) at both ends of a pass and its small maps (latency-bound launches) in the middle, and two streams in
    lockstep are in the same phase at the same time; offset by a fraction of a call, one part's small-map launches run beside the other's
    large-map convs."""
    import threading
    import time
    n = len(fns)
    if n == 1:
        return [fns[0]()]
    device = device if device is not None else torch.cuda.current_device()
    streams = streams or [torch.cuda.Stream(device=device) for _ in range(n)]
    res = [None] * n

    def work(k):
        try:
            torch.cuda.set_device(device)
            if delays is not None and delays[k] > 0:
                time.sleep(delays[k])
            with torch.cuda.stream(streams[k]):
                res[k] = fns[k]()
        except BaseException as e:
            res[k] = e

    th = [threading.Thread(target=work, args=(k,)) for k in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for r in res:
        if isinstance(r, BaseException):
            raise r
    torch.cuda.synchronize(device)
    return res
