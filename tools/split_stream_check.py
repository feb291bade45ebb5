#!/usr/bin/env python3
"""Experiment: does running the two halves of the 64-frame batch on two HIP streams (own handle + workspaces each) hide the
under-filled tail kernels (NMS / top-k / descriptors) of one half under the convolutions of the other?
    python tools/split_stream_check.py [--steps 30]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    from accelerated_features_amd import XFeat
    sd = fixtures.synthetic_state_dict(0)
    xs = [XFeat(weights=sd, top_k=4096) for _ in range(4)]
    x = bench.make_frames(64, seed=1000).cuda()
    streams = [torch.cuda.Stream() for _ in range(4)]

    def one(xf, xb):
        kp, sc, de, nv, nc, cap, hw = xf._detect_device(xb, 4096, 0.05)
        i0, i1, nm = xf.match_pairs_device(de, nv, -1)
        return torch.cat([nv, nc, nm])

    def step_plain():
        return one(xs[0], x).cpu()

    def step_split(k):
        cur = torch.cuda.current_stream()
        outs = []
        n = 64 // k
        for i in range(k):
            streams[i].wait_stream(cur)
            with torch.cuda.stream(streams[i]):
                outs.append(one(xs[i], x[i * n:(i + 1) * n]))
        for i in range(k):
            cur.wait_stream(streams[i])
        return torch.cat(outs).cpu()

    for name, fn in (("plain", step_plain), ("split2", lambda: step_split(2)), ("split4", lambda: step_split(4)),
                     ("plain", step_plain), ("split2", lambda: step_split(2))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        print(f"{name:8s} {dt * 1e3:.4f} ms/step  {64 / dt:9.1f} frames/s")


if __name__ == "__main__":
    main()
