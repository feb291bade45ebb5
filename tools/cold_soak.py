#!/usr/bin/env python3
"""Cold-start torture of the backbone on ONE stream: with xfh_debug_cold_start every matrix-core kernel invalidates the instruction cache when a workgroup starts
(the condition under which the split-bf16 key-point head was found to deliver wrong 16-cell blocks); every output of every step is compared bit for
bit with a reference computed without it.      python tools/cold_soak.py [seconds] [option value ...]      e.g.  cold_soak.py 30 heads_f32 1"""
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
names = ("feats", "heat", "rel", "inv")
with torch.inference_mode():
    f0, _, h0, r0, v0 = xf.net.backbone(x, want_logits=False, want_heat=True, want_invnorm=True)
    want = [t.clone() for t in (f0, h0, r0, v0)]
    torch.cuda.synchronize()
    for cold in (0, 1):
        lib.xfh_debug_cold_start(cold)
        cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
        nbad_vals = torch.zeros(4, dtype=torch.int64, device="cuda")
        t0 = time.time(); steps = 0
        budget = secs if cold else min(secs, 8.0)
        while time.time() - t0 < budget:
            for _ in range(100):
                out = xf.net.backbone(x, want_logits=False, want_heat=True, want_invnorm=True)
                for k, t in enumerate((out[0], out[2], out[3], out[4])):
                    d = (t != want[k])
                    n = d.sum()
                    cnt[k] += (n > 0)
                    nbad_vals[k] += n
                steps += 1
            torch.cuda.synchronize()
        c = cnt.tolist(); nv = nbad_vals.tolist()
        print(f"cold_start {cold} options {opts}: {steps} backbone steps in {time.time() - t0:.1f} s: steps with a differing tensor: " +
              ", ".join(f"{n} {c[k]} ({nv[k]} values)" for k, n in enumerate(names)), flush=True)
lib.xfh_debug_cold_start(0)
