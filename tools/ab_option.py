#!/usr/bin/env python3
"""In-run A/B of a per-handle kernel option on the bench step (same process, same box, alternating rounds):
    python tools/ab_option.py block1 4 5 [span_id]      # span_id: XFH_PROF_* / XFH_SPAN_* id to time (default 3 = block1)
prints the span's mean time and the whole step's time for every value, three rounds each."""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures  # noqa: E402
import bench  # noqa: E402
from accelerated_features_amd import XFeat, _lib  # noqa: E402

key, vals = sys.argv[1], [int(v) for v in sys.argv[2:4]]
span = int(sys.argv[4]) if len(sys.argv) > 4 else 3
xf = XFeat(weights=fixtures.synthetic_state_dict(0), top_k=4096)
x = bench.make_frames(64, seed=1000).cuda()
lib, h = _lib.load(), xf.net.handle()


def step():
    kp, sc, de, nv, nc, cap, hw, d16 = xf._detect_device(x, 4096, 0.05, want_f16=True)
    i0, i1, nm = xf.match_pairs_device(de, nv, -1, d16)
    return nm.cpu()


for rnd in range(3):
    for v in vals:
        xf.set_option(key, v)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        lib.xfh_profile_select(h, _lib.PROF_ALL)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        ids, ms, n = (C.c_int * 4096)(), (C.c_double * 4096)(), C.c_int()
        lib.xfh_profile_read_spans(h, ids, ms, 4096, C.byref(n))
        lib.xfh_profile_select(h, _lib.PROF_NONE)
        tot = sum(ms[i] for i in range(min(n.value, 4096)) if ids[i] == span)
        print(f"round {rnd} {key}={v}: span {span} {1e3 * tot / 5:7.1f} us   step {1e3 * dt:.4f} ms  ({64 / dt:.0f} frames/s)", flush=True)
