"""Multi-GPU path on CPU: one process per 'GPU' over gloo (world size 2).  The data path has no collective (frames shard
embarrassingly); what is checked is the code bench.py itself runs -- accelerated_features_amd.sharding:
shard_range / megadepth shards (disjoint, complete, balanced), sync_barrier + timed_steps (exactly K timed steps,
MAX over ranks), aggregate_rate."""
import os
import socket
import time

import torch
import torch.multiprocessing as mp

from accelerated_features_amd import sharding
from accelerated_features_amd.sharding import shard_range


def test_shard_range_properties():
    for n, w, m in ((64, 8, 2), (64, 3, 2), (10, 4, 1), (2, 4, 2), (1500, 8, 1)):
        seen = []
        for r in range(w):
            b, e = shard_range(n, r, w, m)
            assert 0 <= b <= e <= n and (e - b) % m == 0 and b % m == 0
            seen += list(range(b, e))
        assert seen == list(range(n))
        sizes = [shard_range(n, r, w, m)[1] - shard_range(n, r, w, m)[0] for r in range(w)]
        assert max(sizes) - min(sizes) <= m


def test_megadepth_shards_are_balanced_in_megapixels():
    """BASELINE configs[3] (strong scaling): contiguous chunks of the permuted 1500-pair list; the megapixels per rank --
    the load -- differ by at most 5 % at 2, 4 and 8 ranks."""
    sizes, n_distinct = sharding.megadepth_pair_sizes()
    assert len(sizes) == 1500 and n_distinct >= 10
    assert all(h % 32 == 0 and w % 32 == 0 and max(h, w) <= 1600 for p in sizes for (h, w) in p)
    for world in (1, 2, 4, 8):
        mp_ = sharding.shard_megapixels(sizes, world)
        assert len(mp_) == world and abs(sum(mp_) - sharding.shard_megapixels(sizes, 1)[0]) < 1e-6
        assert (max(mp_) - min(mp_)) / (sum(mp_) / world) <= 0.05, (world, mp_)


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_PORT"] = str(port)
    dist = sharding.init_process_group("gloo", rank, world)            # 127.0.0.1 rendezvous, like bench.py
    b, e = shard_range(n_frames, rank, world, multiple=2)
    owned = torch.zeros(n_frames, dtype=torch.int32)
    owned[b:e] = 1
    frames = torch.arange(n_frames, dtype=torch.float64)
    calls = {"n": 0, "armed": 0}

    def step():                                     # "process" the shard; rank 1 is slower
        calls["n"] += 1
        time.sleep(0.02 * (rank + 1))
        return (frames[b:e] ** 2).sum()

    def arm(_):
        calls["armed"] = calls["n"]

    secs, last = sharding.timed_steps(step, steps=3, warmup=2, dist=dist, device_sync=None, device="cpu", before_timed=arm)
    dist.all_reduce(owned, op=dist.ReduceOp.SUM)
    tot = last.clone()
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    rate = sharding.aggregate_rate(e - b, 3, world, secs, "weak")
    q.put((rank, owned.tolist(), secs, float(tot), calls["n"], calls["armed"], rate))
    sharding.sync_barrier(dist)
    dist.destroy_process_group()


def test_two_process_gloo_sharding_and_timing_protocol():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n = 64
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    secs = [r[2] for r in res]
    assert secs[0] == secs[1]                        # both ranks hold the MAX over ranks
    assert 3 * 0.04 <= secs[0] < 3 * 0.04 + 0.5      # = the slow rank's three timed steps (rank 0 alone would be 0.06 s)
    for rank, owned, tmax, tot, n_calls, armed, rate in res:
        assert owned == [1] * n                      # disjoint and complete
        assert tot == float(sum(i * i for i in range(n)))
        assert n_calls == 5 and armed == 2           # exactly warmup + steps calls; the hook ran between them
        assert abs(rate - 32 * 3 * 2 / tmax) < 1e-9  # whole-job units / max time


def _host_worker(rank, world, port, q):
    os.environ["MASTER_PORT"] = str(port)
    g = sharding.HostGroup(rank, world, token=f"pytest_{port}")
    secs, last = sharding.timed_steps(lambda: time.sleep(0.02 * (rank + 1)) or rank, steps=3, warmup=1, dist=g, device_sync=None, device="cpu")
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    g.all_reduce(t, op=g.ReduceOp.SUM)
    q.put((rank, secs, float(t.item())))
    g.destroy_process_group()


def test_host_group_is_a_barrier_and_a_max_without_a_collective_library():
    """sharding.HostGroup (files under /dev/shm): timed_steps' bracket and max-over-ranks through it, three ranks"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = sharding.free_port()
    procs = [ctx.Process(target=_host_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert len({r[1] for r in res}) == 1 and 3 * 0.06 <= res[0][1] < 3 * 0.06 + 0.5      # every rank holds the slowest rank's time
    assert all(r[2] == 6.0 for r in res)


def test_bench_launches_its_own_ranks_and_never_underreports(tmp_path):
    """`python bench.py --gpus 2` WITHOUT a launcher re-runs itself as two ranks under torch.distributed.run (it used to run one
    rank and print n_gpus = 1).  --launch-selftest swaps RCCL / the hot path for gloo / a sleep so that the launch path itself runs here;
    without it, on a box with fewer GPUs than asked for, the command must refuse loudly instead of printing a line."""
    import json
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-selftest", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]                      # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["selftest"] is True
    assert d["ms_per_step"] >= 19.0                               # the slow rank (20 ms sleeps) sets the time: MAX over ranks
    # the same at the width of a full node: 8 ranks, one line, n_gpus 8, the slowest rank's time
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--launch-selftest", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 3 and d["selftest"] is True and d["scaling"] == "weak"
    assert d["ms_per_step"] >= 79.0                               # rank 7 sleeps 80 ms per step: MAX over ranks
    assert abs(d["value"] - 8 * d["config"]["units_per_rank_per_step"] / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]      # whole-job units / the slowest rank's time
    assert d["barrier"] == "gloo"
    # the same launch with the host-side file barrier (no collective library at all) ...
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--launch-selftest", "--steps", "3", "--warmup", "1", "--barrier", "host"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["barrier"].startswith("file (requested") and 39.0 <= d["ms_per_step"] < 60.0      # rank 3's 40 ms sleeps: MAX over ranks through the files
    # ... and the fallback: the collective's bring-up fails on ONE rank (test hook; the others then fail or time out in the rendezvous that rank never joins) -> every
    # rank agrees on the file barrier, the line still comes out, rc 0
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--launch-selftest", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env={**env, "XFH_TEST_FAIL_COLLECTIVE": "1"}, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and re.match(r"file \(gloo bring-up failed on [1-3] of 3 ranks", d["barrier"]) and 29.0 <= d["ms_per_step"] < 50.0
    # the real workload on this GPU-less box: refuse, do not report
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return                                                    # (a multi-GPU box would really run it)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "refusing to run" in (r.stderr + r.stdout)
    # a launcher whose world size disagrees with --gpus is an error too
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--launch-selftest"], capture_output=True, text=True, timeout=120,
                       env={**env, "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, cwd=str(tmp_path))
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
