#!/usr/bin/env python3
"""In-run A/B of option SETS on the bench step (one process, one box, alternating rounds):
    python tools/ab_configs.py "block1=0" "block1=6" "block1=7" --spans 3
    python tools/ab_configs.py "heads_f32=2" "heads_f32=0,fx=11" --spans 202,203
Every configuration: the listed spans' mean time (XFH_PROF_* / XFH_SPAN_* ids of include/xfeat_hip.h; default 3 = block1) and the whole step's time, three rounds each;
the first configuration's results are the reference the others' key-point lists are compared with (equal counts, coordinates within 0.01 px on >= 99.9 % of the rows)."""
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

args = [a for a in sys.argv[1:] if not a.startswith("--")]
spans = [3]
if "--spans" in sys.argv:
    spans = [int(v) for v in sys.argv[sys.argv.index("--spans") + 1].split(",")]
    args.remove(sys.argv[sys.argv.index("--spans") + 1])
configs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in a.split(",")) for a in args]
keys = sorted({k for c in configs for k in c})
sd = fixtures.synthetic_state_dict(0)
x = bench.make_frames(64, seed=1000).cuda()
lib = _lib.load()
models = []
for c in configs:
    m = XFeat(weights=sd, top_k=4096)
    for k, v in c.items():
        m.set_option(k, v)
    models.append(m)


def step(xf):
    kp, sc, de, nv, nc, cap, hw, d16 = xf._detect_device(x, 4096, 0.05, want_f16=True)
    i0, i1, nm = xf.match_pairs_device(de, nv, -1, d16)
    return kp, nv, nm.cpu()


ref = None
for c, m in zip(configs, models):
    kp, nv, nm = step(m)
    st = m.net.take_status()
    if ref is None:
        ref = (kp.clone(), nv.clone())
        print(f"{c}: reference ({int(nv.sum())} key-points), status {st}")
    else:
        same = bool(torch.equal(nv, ref[1]))
        close = float(((kp - ref[0]).abs().amax(-1) <= 0.01).float().mean()) if same else 0.0
        print(f"{c}: counts equal {same}, rows within 0.01 px {100 * close:.3f} %, status {st}")
for rnd in range(3):
    for c, m in zip(configs, models):
        h = m.net.handle()
        for _ in range(3):
            step(m)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            step(m)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        lib.xfh_profile_select(h, _lib.PROF_ALL)
        for _ in range(5):
            step(m)
        torch.cuda.synchronize()
        ids, ms, n = (C.c_int * 4096)(), (C.c_double * 4096)(), C.c_int()
        lib.xfh_profile_read_spans(h, ids, ms, 4096, C.byref(n))
        lib.xfh_profile_select(h, _lib.PROF_NONE)
        per = {s: 1e3 * sum(ms[i] for i in range(min(n.value, 4096)) if ids[i] == s) / 5 for s in spans}
        print(f"round {rnd} {c}: " + "  ".join(f"span {s} {v:7.1f} us" for s, v in per.items()) + f"   step {1e3 * dt:.4f} ms  ({64 / dt:.0f} frames/s)", flush=True)
