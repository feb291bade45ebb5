#!/usr/bin/env python3
"""Two-lane throughput (FrameStream) for two values of a per-handle option, alternating rounds:  python tools/lanes_option_ab.py block1 4 5"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures, bench
from accelerated_features_amd import XFeat
from accelerated_features_amd.streaming import FrameStream
key, vals = sys.argv[1], [int(v) for v in sys.argv[2:4]]
B = 64
xfs = [XFeat(weights=fixtures.synthetic_state_dict(0), top_k=4096) for _ in range(2)]
x = bench.make_frames(B, seed=1000).cuda()
fs = FrameStream(xfeats=xfs, top_k=4096)


def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        if fs.in_flight == fs.lanes: fs.result()
        fs.submit(x)
    fs.drain(); torch.cuda.synchronize()
    return B * n / (time.perf_counter() - t0)


run(150)
for rnd in range(4):
    out = []
    for v in vals:
        for xf in xfs: xf.set_option(key, v)
        run(10)
        out.append(f"{key}={v}: {run(60):.0f}")
    print(f"round {rnd}: " + "   ".join(out), flush=True)
