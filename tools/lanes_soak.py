#!/usr/bin/env python3
"""Soak of the two-lane path at the bench shape: N batches two at a time, every retired result compared bit for bit with the synchronous result.
    python tools/lanes_soak.py [steps] [key value ...]      # e.g. tools/lanes_soak.py 600 bx 5"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from accelerated_features_amd import XFeat
from accelerated_features_amd.streaming import FrameStream
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
opts = list(zip(sys.argv[2::2], [int(v) for v in sys.argv[3::2]]))
sd = fixtures.synthetic_state_dict(0)
xfs = [XFeat(weights=sd, top_k=4096) for _ in range(3)]
for xf in xfs:
    for k, v in opts: xf.set_option(k, v)
x = torch.cat([fixtures.texture_images(8, 480, 640, seed=77)] * 8).cuda()
ref = xfs[2]
kp0, sc0, de0, nv0, nc0, cap, hw, d16 = ref._detect_device(x, 4096, 0.05, want_f16=True)
i00, i10, nm0 = ref.match_pairs_device(de0, nv0, -1, d16)
nv_h, nm_h = nv0.cpu(), nm0.cpu()
fs = FrameStream(xfeats=xfs[:2], top_k=4096)
bad = {"counts": 0, "kp": 0, "scores": 0, "desc": 0, "idx": 0}


def check(r, step):
    ok = True
    if not (torch.equal(r["n_valid"], nv_h) and torch.equal(r["n_matches"], nm_h)): bad["counts"] += 1; ok = False
    if not torch.equal(r["keypoints"], kp0): bad["kp"] += 1; ok = False
    if not torch.equal(r["scores"], sc0): bad["scores"] += 1; ok = False
    if not torch.equal(r["descriptors"], de0):
        bad["desc"] += 1; ok = False
        d = (r["descriptors"] - de0).abs()
        print(f"  step {step}: descriptors differ: max {float(d.max()):.3g}, {int((d > 0).sum())} values in images {sorted(set((d.flatten(1).max(1).values > 0).nonzero().flatten().tolist()))}", flush=True)
    if ok:
        for p in range(32):
            n = int(nm_h[p])
            if not (torch.equal(r["idx0"][p, :n], i00[p, :n]) and torch.equal(r["idx1"][p, :n], i10[p, :n])): bad["idx"] += 1; break


for step in range(steps):
    if fs.in_flight == fs.lanes: check(fs.result(), step)
    fs.submit(x)
for r in fs.drain(): check(r, steps)
print(f"{steps} steps, options {opts}: mismatching results {bad}")
