#!/usr/bin/env python3
"""Per-kernel time of xfh_match_mnn at the bench shape (32 pairs of 4096 x 4096 unit descriptors from the bench frames), HIP-event spans.
    python tools/match_time.py            # XFH_LIB_PATH=<variant .so> selects another build of the library
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures  # noqa: E402
import bench  # noqa: E402
from accelerated_features_amd import XFeat, _lib  # noqa: E402

NAMES = {220: "zero", 221: "prep", 222: "sweep", 223: "refine", 224: "finalize", 225: "exact"}
xf = XFeat(weights=fixtures.synthetic_state_dict(0), top_k=4096)
x = bench.make_frames(64, seed=1000).cuda()
kp, sc, de, nv, nc, cap, hw, d16 = xf._detect_device(x, 4096, 0.05, want_f16=True)
lib, h = _lib.load(), xf.net.handle()
for prepared in (True, False):
    for _ in range(3):
        xf.match_pairs_device(de, nv, -1, d16 if prepared else None)
    lib.xfh_profile_select(h, _lib.PROF_ALL)
    n_it = 10
    for _ in range(n_it):
        i0, i1, nm = xf.match_pairs_device(de, nv, -1, d16 if prepared else None)
    torch.cuda.synchronize()
    ids = (C.c_int * 4096)(); ms = (C.c_double * 4096)(); n = C.c_int()
    lib.xfh_profile_read_spans(h, ids, ms, 4096, C.byref(n))
    lib.xfh_profile_select(h, _lib.PROF_NONE)
    acc = {}
    for i in range(min(n.value, 4096)):
        acc[ids[i]] = acc.get(ids[i], 0.0) + ms[i]
    print(("prepared fp16 copies" if prepared else "own conversion      "), os.environ.get("XFH_LIB_PATH", "default"),
          " ".join(f"{NAMES.get(k, k)} {1e3 * v / n_it:7.1f} us" for k, v in sorted(acc.items())), f"| total {1e3 * sum(acc.values()) / n_it:7.1f} us | matches {int(nm.sum())}", flush=True)
