"""Frame sharding and the timing protocol for multi-GPU runs.

Frames (and frame pairs) are independent and the model is 6 MB, so every GPU holds a replica
and processes a contiguous chunk of the work list: no data-path collective exists.  One
process per GPU; torch.distributed (RCCL on the GPUs, gloo in the CPU tests) is used only for
the start/stop barrier and the max-over-ranks timing of the benchmark -- with a host-side file
barrier (HostGroup) as the fallback when the collective library does not come up (open_group).  `bench.py` and
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


class HostGroup:
    """The subset of torch.distributed the benchmark needs -- barrier(), all_reduce(t, op=MAX | SUM) on a host tensor, destroy_process_group() -- WITHOUT a collective
    library: the ranks of ONE node meet in files under /dev/shm (memory-backed; the temp directory elsewhere).  The data path has no collective, so RCCL is only
    ever the start / stop barrier and the max-over-ranks time of bench.py: this class is what `bench.py --barrier host` uses, and what a failed RCCL bring-up falls back to
    (open_group), so that an environment hiccup in the first collective of an 8-GPU run costs a note in the line instead of the scaling record.
    Protocol: operation k of rank r publishes `<dir>/<k>.<r>` (its value, written to a temp name and renamed: readers never see half a file), then polls until all `world`
    files of operation k exist; values are 8-byte doubles.  Rank 0 removes operation k - 2's files (every rank has left k - 1, hence finished reading k - 2)."""
    name = "file"

    class ReduceOp:
        MAX, SUM = "max", "sum"

    def __init__(self, rank, world, token=None, timeout_s=600.0):
        import tempfile
        self.rank, self.world, self.k, self.timeout_s = rank, world, 0, timeout_s
        # one directory per launch: the rendezvous port and the launcher's pid (the parent of every rank) -- a crashed earlier run on the same port leaves nothing to trip over
        token = token or f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
        self.dir = os.path.join(base, f"xfh_hostgroup_{os.getuid()}_{token}")
        os.makedirs(self.dir, exist_ok=True)

    def _exchange(self, value):
        import struct
        k = self.k
        self.k += 1
        tmp = os.path.join(self.dir, f".{k}.{self.rank}.tmp")
        with open(tmp, "wb") as f:
            f.write(struct.pack("<d", float(value)))
        os.rename(tmp, os.path.join(self.dir, f"{k}.{self.rank}"))
        vals, t0 = [None] * self.world, time.perf_counter()
        while True:
            for r in range(self.world):
                if vals[r] is None:
                    try:
                        with open(os.path.join(self.dir, f"{k}.{r}"), "rb") as f:
                            b = f.read()
                        if len(b) == 8:
                            vals[r] = struct.unpack("<d", b)[0]
                    except FileNotFoundError:
                        pass
            if all(v is not None for v in vals):
                break
            dt = time.perf_counter() - t0
            if dt > self.timeout_s:
                raise RuntimeError(f"HostGroup: rank {self.rank} waited {dt:.0f} s for operation {k}: ranks {[r for r in range(self.world) if vals[r] is None]} never arrived")
            if dt > 0.05:
                time.sleep(0.0002)           # (the first 50 ms spin: a barrier's skew is what the timed region sees)
        if self.rank == 0 and k >= 2:
            for r in range(self.world):
                try:
                    os.remove(os.path.join(self.dir, f"{k - 2}.{r}"))
                except OSError:
                    pass
        return vals

    def barrier(self):
        self._exchange(0.0)

    def all_reduce(self, t, op="max"):
        vals = self._exchange(float(t.reshape(-1)[0].item()))
        t.reshape(-1)[0] = max(vals) if op == "max" else sum(vals)
        return t

    def get_backend(self):
        return self.name

    def destroy_process_group(self):
        self.barrier()
        # leaving the barrier means every rank has PUBLISHED its file, not that every rank has read them all: each rank marks the end of its reading, and rank 0 removes
        # the directory only when all marks are there (a rank still polling in a directory that vanished would wait out its timeout)
        open(os.path.join(self.dir, f"done.{self.rank}"), "wb").close()
        if self.rank == 0:
            import shutil
            t0 = time.perf_counter()
            while not all(os.path.exists(os.path.join(self.dir, f"done.{r}")) for r in range(self.world)) and time.perf_counter() - t0 < self.timeout_s:
                time.sleep(0.0005)
            shutil.rmtree(self.dir, ignore_errors=True)


def open_group(backend, rank, world, device=None, probe_timeout_s=120.0, force_host=False, fail_probe=False):
    """The rank group of a benchmark run: (group, note).  `backend` ("nccl" = RCCL on the GPUs, "gloo" in the CPU self-tests) is brought up and PROBED -- one all_reduce on
    `device`, in a watchdog thread with a timeout, because RCCL creates its communicator lazily in the first collective and a failure there is a hang as often as an
    exception -- and the ranks then agree THROUGH FILES (HostGroup) whether every one of them got through.  All fine: the torch.distributed module is returned (note None).
    Any rank failed, timed out, or force_host: every rank uses the HostGroup (note says why), so the run still prints its line.  fail_probe: test hook (this rank's probe raises)."""
    import threading
    import torch
    host = HostGroup(rank, world)
    if force_host:
        host.barrier()
        return host, "requested (--barrier host)"
    outcome = {"ok": False, "err": None}
    dist_mod = [None]

    def bring_up():
        try:
            if fail_probe:
                raise RuntimeError("probe failure injected by the test hook")
            import datetime
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=probe_timeout_s))
            dist_mod[0] = dist
            t = torch.ones(1, dtype=torch.float32, device=device if device is not None else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            if t.is_cuda:
                torch.cuda.synchronize()
            outcome["ok"] = abs(float(t.item()) - world) < 0.5
            if not outcome["ok"]:
                outcome["err"] = f"probe all_reduce returned {float(t.item())}, expected {world}"
        except Exception as e:              # noqa: BLE001 -- whatever the collective library throws
            outcome["err"] = f"{type(e).__name__}: {e}"
    th = threading.Thread(target=bring_up, daemon=True)
    th.start()
    th.join(probe_timeout_s)
    if th.is_alive():
        outcome["err"] = f"{backend} bring-up did not finish in {probe_timeout_s:.0f} s"
    t = torch.tensor([0.0 if outcome["ok"] else 1.0], dtype=torch.float64)
    host.all_reduce(t, op=HostGroup.ReduceOp.SUM)
    n_bad = int(t.item())
    if n_bad == 0:
        host.destroy_process_group()
        return dist_mod[0], None
    note = f"{backend} bring-up failed on {n_bad} of {world} ranks" + (f" (this rank: {outcome['err']})" if outcome["err"] else "") + ": host-side file barrier instead"
    host.abandoned_backend = th.is_alive() or dist_mod[0] is not None      # (a half-initialised collective library: leave the process with os._exit)
    return host, note


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
    t = torch.tensor([dt], dtype=torch.float64, device="cpu" if isinstance(dist, HostGroup) else device)
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
