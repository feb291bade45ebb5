#!/usr/bin/env python3
"""LighterGlue timing on one MI355X: one pair of N x N key-points (default 4096 = the reference's top_k), synthetic
weights and inputs (tests/fixtures.py).  Prints ms/pair for the HIP path and, with --cpu, the torch-CPU oracle.
    python tools/bench_lighterglue.py [--n 4096] [--iters 20] [--prune 1536] [--cpu]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--prune", type=int, default=1536)
    ap.add_argument("--cpu", action="store_true")
    a = ap.parse_args()
    from accelerated_features_amd.lighterglue import LighterGlue
    sd = fixtures.lighterglue_state_dict(0)
    lg = LighterGlue(weights=sd)
    k0, d0, s0, k1, d1, s1 = fixtures.lighterglue_inputs(a.n, a.n, seed=1)
    args = (k0.cuda(), d0.cuda(), s0.tolist(), k1.cuda(), d1.cuda(), s1.tolist(), 0.1, a.prune)
    for _ in range(3):
        m, s, c = lg.match_device(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        m, s, c = lg.match_device(*args)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f"hip: N={a.n} prune>{a.prune}: {ms:.3f} ms/pair, {int(c.item())} matches")
    if a.cpu:
        from oracle import lighterglue_oracle as LG
        t = time.perf_counter()
        rm, rs = LG.lighterglue_forward(sd, k0, d0, s0, k1, d1, s1, min_conf=0.1, prune=True, prune_min_kpts=a.prune)
        dt = time.perf_counter() - t
        print(f"cpu oracle ({torch.get_num_threads()} threads): {dt * 1e3:.1f} ms/pair, {len(rm)} matches")


if __name__ == "__main__":
    main()
