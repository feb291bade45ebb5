"""Frame sharding and the timing protocol for multi-GPU runs.

Frames (and frame pairs) are independent and the model is 6 MB, so every GPU holds a replica
and processes a contiguous chunk of the work list: no data-path collective exists.  One
process per GPU; torch.distributed (RCCL on the GPUs, gloo in the CPU tests) is used only for
the start/stop barrier and the max-over-ranks timing of the benchmark.  `bench.py` and
`tests/test_sharding_gloo.py` both run THIS code.
"""
import json
import os
import time

import numpy as np


def shard_range(n_items, rank, world, multiple=1):
    """Contiguous [begin, end) of `n_items` work units for `rank` of `world`.

    Units are spread as evenly as possible (sizes differ by at most `multiple`); `multiple`
    keeps chunk boundaries on multiples of that many units (2 for frame pairs)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    if n_items % multiple:
        raise ValueError(f"{n_items} items is not a multiple of {multiple}")
    groups = n_items // multiple
    base, extra = divmod(groups, world)
    begin = rank * base + min(rank, extra)
    end = begin + base + (1 if rank < extra else 0)
    return begin * multiple, end * multiple


def rank_world():
    """(rank, local_rank, world) from the torch.distributed.run environment (1 process per GPU)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(script, argv, n_ranks, visible_devices=None, python=None):
    """`python bench.py --gpus N` started WITHOUT a launcher (WORLD_SIZE unset): re-run the same command line as N ranks under
    torch.distributed.run (one process per GPU, 127.0.0.1 rendezvous on a free port) and return its exit code.  Refuses -- loudly --
    when fewer than N devices are visible, so a line with n_gpus != the request can never be printed.
    visible_devices: the device count to check against (None: do not check -- CPU / gloo self-tests)."""
    import subprocess
    import sys
    if "WORLD_SIZE" in os.environ:
        raise RuntimeError("self_launch called inside a torch.distributed.run worker")
    if visible_devices is not None and visible_devices < n_ranks:
        raise SystemExit(f"--gpus {n_ranks} requested but only {visible_devices} GPU(s) are visible to this process: refusing to run "
                         f"(a benchmark line must never report fewer GPUs than were asked for)")
    cmd = [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs on these hosts
    return subprocess.call(cmd, env=env)


def init_process_group(backend, rank, world):
    """Rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def sync_barrier(dist=None, device_sync=None):
    """device sync, barrier over the ranks, device sync: the bracket the benchmark contract prescribes."""
    if device_sync is not None:
        device_sync()
    if dist is not None:
        dist.barrier()
    if device_sync is not None:
        device_sync()


def timed_steps(step, steps, warmup, dist=None, device_sync=None, device="cpu", before_timed=None):
    """`warmup` untimed calls of step(), then EXACTLY `steps` timed calls bracketed by sync_barrier on both sides.
    Returns (seconds = MAX over ranks of the local elapsed time, result of the last step).  `before_timed` runs after the
    warm-up, outside the timed region (profiler arming)."""
    import torch
    last = None
    for _ in range(warmup):
        last = step()
    if before_timed is not None:
        before_timed(last)
    sync_barrier(dist, device_sync)
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    sync_barrier(dist, device_sync)
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item()), last


def aggregate_rate(units_per_rank_step, steps, world, seconds, scaling="weak"):
    """Whole-job throughput: weak scaling = every rank processed `units_per_rank_step` per step; strong = the job's units
    (`units_per_rank_step` = the whole list) were split across the ranks."""
    total = units_per_rank_step * steps * (world if scaling == "weak" else 1)
    return total / seconds


# ------------------------------------------------------------------------------------------------------------------
# BASELINE configs[3]: the MegaDepth-1500 pair list (sizes only), long side 1600, sharded contiguously
# ------------------------------------------------------------------------------------------------------------------
_SIZES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "megadepth1500_sizes.json")


def megadepth_pair_sizes(path=_SIZES, long_side=1600, seed=15):
    """[((h0,w0),(h1,w1)), ...] for the 1500 pairs of the reference's assets/megadepth_1500.json (sizes committed as
    accelerated_features_amd/data/megadepth1500_sizes.json by tests/golden/make_megadepth_sizes.py), scaled x long_side/1184 and floored to multiples of 32, in a fixed
    pseudo-random order (the dataset interleaves scenes; a permuted list also balances the contiguous shards)."""
    rows = json.load(open(path))
    up = lambda v: max(32, int(v * long_side / 1184) // 32 * 32)
    sizes = []
    for h0, w0, h1, w1, n in rows:
        sizes += [((up(h0), up(w0)), (up(h1), up(w1)))] * n
    order = np.random.RandomState(seed).permutation(len(sizes))
    return [sizes[i] for i in order], len(rows)


def shard_megapixels(sizes, world):
    """megapixels each rank's contiguous shard holds (the load the shard represents)."""
    out = []
    for r in range(world):
        lo, hi = shard_range(len(sizes), r, world)
        out.append(sum(a[0] * a[1] + b[0] * b[1] for a, b in sizes[lo:hi]) / 1e6)
    return out
