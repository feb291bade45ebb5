#!/usr/bin/env python3
"""Calibration of bench.py's `cpu_baseline` ("kind": "port"): the UNMODIFIED reference (/root/reference/modules/xfeat.py, CPU) and the
oracle port timed on the same cores, same inputs, same workload shape (batches of 4 VGA frames: detectAndCompute top_k=4096 + the 2 pair
matches).  Container only -- /root/reference does not exist on the GPU box, which is why bench.py times the port there.

    python tools/cpu_reference_vs_oracle.py [seconds-per-side]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

import fixtures  # noqa: E402
from bench import make_frames, TOP_K  # noqa: E402
from modules.xfeat import XFeat as RefXFeat  # noqa: E402  (the reference, unmodified)
from oracle import xfeat_oracle as O  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
sd = fixtures.synthetic_state_dict(0)
x = make_frames(4, seed=77)
ref = RefXFeat(weights=sd, top_k=TOP_K, detection_threshold=0.05)
assert ref.dev.type == "cpu"
torch.set_num_threads(os.cpu_count() or 1)


def ref_batch():
    out = ref.detectAndCompute(x, top_k=TOP_K)
    for p in range(2):
        ref.match(out[2 * p]["descriptors"], out[2 * p + 1]["descriptors"], -1)


def oracle_batch():
    out = O.detect_and_compute(sd, x, top_k=TOP_K)
    for p in range(2):
        O.match_mnn(out[2 * p]["descriptors"], out[2 * p + 1]["descriptors"], -1)


for name, fn in (("reference", ref_batch), ("oracle port", oracle_batch)):
    fn()
    n, used = 0, 0.0
    while used < seconds:
        t0 = time.perf_counter()
        fn()
        used += time.perf_counter() - t0
        n += 1
    print(f"{name:12s}: {4 * n / used:7.2f} frames/s ({n} batches of 4 VGA frames in {used:.1f} s, {torch.get_num_threads()} threads)")
