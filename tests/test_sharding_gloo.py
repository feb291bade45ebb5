"""Multi-GPU path on CPU: one process per 'GPU' over gloo (world size 2).  The data path has no
collective (frames shard embarrassingly); what is checked is that the shards are
disjoint, cover every frame pair, and that the benchmark's barrier + max-over-ranks timing
protocol works."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

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


def _worker(rank, world, port, n_frames, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = shard_range(n_frames, rank, world, multiple=2)
    owned = torch.zeros(n_frames, dtype=torch.int32)
    owned[b:e] = 1
    # "process" the shard: a per-frame checksum that only the owner computes
    frames = torch.arange(n_frames, dtype=torch.float64)
    local = (frames[b:e] ** 2).sum()
    dist.barrier()
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)      # pretend per-rank elapsed time
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(owned, op=dist.ReduceOp.SUM)
    tot = local.clone()
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    q.put((rank, owned.tolist(), float(t), float(tot)))
    dist.barrier()
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
    for rank, owned, tmax, tot in res:
        assert owned == [1] * n                      # disjoint and complete
        assert abs(tmax - 0.2) < 1e-12               # max over ranks
        assert tot == float(sum(i * i for i in range(n)))
