#!/usr/bin/env python3
"""Cold-start torture of the whole step on ONE stream: with xfh_debug_cold_start every matrix-core kernel invalidates the instruction cache when a workgroup starts
(the condition under which the split-bf16 key-point head was found to deliver wrong 16-cell blocks); every output of every step -- network outputs,
key-points, descriptors, match lists -- is compared bit for bit with a reference computed without it.  With a library built by `build.py --shift N` (XFH_LIB_PATH) the
kernels' code sits 4 N bytes further along the instruction-cache lines: tools/shift_scan.sh runs this for N = 1 .. 15.
    python tools/cold_soak.py [seconds] [option value ...]      e.g.  cold_soak.py 30 heads_f32 1"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from accelerated_features_amd import XFeat, _lib
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
opts = list(zip(sys.argv[2::2], [int(v) for v in sys.argv[3::2]]))
lib = _lib.load()
sd = fixtures.synthetic_state_dict(0)
x = torch.cat([fixtures.texture_images(8, 480, 640, seed=77)] * 8).cuda()
xf = XFeat(weights=sd, top_k=4096)
for k_, v_ in opts: xf.set_option(k_, v_)
names = ("feats", "heat", "rel", "inv", "kpts", "desc", "idx0", "idx1", "counts")


def step():
    f, _, h, r, inv = xf.net.backbone(x, want_logits=False, want_heat=True, want_invnorm=True)
    kp, sc, de, nv, nc, cap, hw, d16 = xf._detect_device(x, 4096, 0.05, want_f16=True)
    i0, i1, nm = xf.match_pairs_device(de, nv, -1, d16)
    return f, h, r, inv, kp, de, i0, i1, torch.cat([nv, nc, nm])


with torch.inference_mode():
    want = [t.clone() for t in step()]
    lo_k, lo_m = int(want[8][:64].min()), int(want[8][128:].min())
    torch.cuda.synchronize()
    rc = 0
    for cold in (0, 1):
        lib.xfh_debug_cold_start(cold)
        bad = torch.zeros(len(names), dtype=torch.int64, device="cuda")
        t0 = time.time(); n = 0
        budget = secs if cold else min(secs, 4.0)
        while time.time() - t0 < budget:
            for _ in range(50):
                for k, t in enumerate(step()):
                    w = want[k]
                    if k in (4, 5): d = (t[:, :lo_k] != w[:, :lo_k]).any()
                    elif k in (6, 7): d = (t[:, :lo_m] != w[:, :lo_m]).any()
                    else: d = (t != w).any()
                    bad[k] += d
                n += 1
            torch.cuda.synchronize()
        b = bad.tolist()
        rc |= int(any(b))
        print(f"{os.path.basename(_lib.LIB_PATH)} cold_start {cold} options {opts}: {n} steps in {time.time() - t0:.1f} s: steps with a differing tensor: " +
              ", ".join(f"{nm_} {b[k]}" for k, nm_ in enumerate(names)), flush=True)
lib.xfh_debug_cold_start(0)
sys.exit(rc)
