#!/usr/bin/env python3
"""Two batches in flight on two HIP streams (two handles = two workspaces) against one stream: does the chip fill the latency-bound tail of one
step (NMS compaction, top-k, refine scan, finalize: ~120 us at low occupancy) with the convolutions of the next?"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures, bench
from accelerated_features_amd import XFeat
B = 64
xfs = [XFeat(weights=fixtures.synthetic_state_dict(0), top_k=4096) for _ in range(4)]
xs = [bench.make_frames(B, seed=1000).cuda() for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]
host = [torch.empty((3, B), dtype=torch.int32).pin_memory() for _ in range(4)]
dev = [torch.zeros((3, B), dtype=torch.int32, device="cuda") for _ in range(4)]
ev = [torch.cuda.Event() for _ in range(4)]


def queue(i, k):          # step i on lane k
    with torch.cuda.stream(streams[k]):
        xf, d = xfs[k], dev[k]
        kp, sc, de, nv, nc, cap_, hw, d16 = xf._detect_device(xs[k], 4096, 0.05, want_f16=True, counts_out=d[:2])
        xf.match_pairs_device(de, nv, -1, d16, n_out=d[2, :B // 2])
        host[k].copy_(d, non_blocking=True)
        ev[k].record(streams[k])


def run(n, lanes):
    for k in range(lanes):
        queue(0, k); ev[k].synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        k = i % lanes
        if i >= lanes: ev[k].synchronize()        # the lane's previous step has delivered its counts
        queue(i, k)
    for k in range(lanes): ev[k].synchronize()
    torch.cuda.synchronize()
    return B * n / (time.perf_counter() - t0)


for rnd in range(3):
    print(f"round {rnd}: lanes 1 {run(48, 1):9.0f}   2 {run(48, 2):9.0f}   3 {run(48, 3):9.0f}   4 {run(48, 4):9.0f} frames/s", flush=True)
