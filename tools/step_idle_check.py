#!/usr/bin/env python3
"""How long is the GPU idle between two sparse steps because of the per-step read-back of the counts?
    python tools/step_idle_check.py"""
import sys, time, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fixtures, bench
from accelerated_features_amd import XFeat
xf = XFeat(weights=fixtures.synthetic_state_dict(0), top_k=4096)
x = bench.make_frames(64, seed=1000).cuda()
def launch():
    kp, sc, de, nv, nc, cap, hw = xf._detect_device(x, 4096, 0.05)
    i0, i1, nm = xf.match_pairs_device(de, nv, -1)
    return torch.cat([nv, nc, nm])
for mode in ("sync", "nosync", "sync", "nosync", "pipelined"):
    for _ in range(5): launch().cpu()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    prev = None
    for _ in range(40):
        c = launch()
        if mode == "sync": c.cpu()
        elif mode == "pipelined":
            if prev is not None: prev.cpu()        # read step i-1 while step i runs
            prev = c
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
    print(f"{mode:10s} {dt*1e3:.4f} ms/step {64/dt:.1f} fps")
