#!/usr/bin/env python3
"""Time xfh_find_homography (three launches) on synthetic match lists: the demo's shape (one pair, ~1000 matches) and the bench batch
(32 pairs x 4096 rows).   python tools/homography_time.py [P,n,maxIters]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from accelerated_features_amd.homography import find_homography_batch  # noqa: E402
from test_oracle_homography import synthetic_pair  # noqa: E402

CASES = ((1, 300, 700), (1, 1000, 700), (1, 4096, 700), (32, 1024, 700), (32, 4096, 700), (32, 4096, 4096))
if len(sys.argv) > 1:                      # one case "P,n,iters" (per-kernel profiles: rocprofv3 --kernel-trace --stats -- python tools/homography_time.py 1,1000,700)
    CASES = (tuple(int(v) for v in sys.argv[1].split(",")),)
for P, n, iters in CASES:
    p0 = np.zeros((P, n, 2), np.float32)
    p1 = np.zeros_like(p0)
    for p in range(P):
        p0[p], p1[p], _, _ = synthetic_pair(n, 0.5, 0.7, seed=p)
    a, b = torch.from_numpy(p0).cuda(), torch.from_numpy(p1).cuda()
    for _ in range(3):
        r = find_homography_batch(a, b, None, 4.0, iters, 0.995, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        r = find_homography_batch(a, b, None, 4.0, iters, 0.995, 1)
    e1.record()
    torch.cuda.synchronize()
    info = r["info"].cpu().numpy()
    print(f"P {P:3d} n {n:5d} maxIters {iters:4d}: {e0.elapsed_time(e1) / reps * 1e3:8.1f} us per call, found {int(info[:, 0].sum())}/{P}, "
          f"mean inliers {info[:, 3].mean():.0f}, loop iterations {info[:, 2].mean():.0f}", flush=True)
