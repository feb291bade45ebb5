#!/usr/bin/env python3
"""Locate run-to-run nondeterminism in the semi-dense path: run every stage twice, compare bitwise."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from accelerated_features_amd import XFeat
xf = XFeat(weights=fixtures.synthetic_state_dict(0))
base = fixtures.texture_images(2, 1024, 1024, seed=55)
a = torch.cat([base, base.flip(3), base.flip(2), base.flip(2).flip(3)]).cuda()
b = torch.roll(a, (16, 24), (2, 3)).contiguous()
for rep in range(3):
    d1 = [xf.detectAndComputeDense(a, top_k=4096) for _ in range(2)]
    print("dense", {k: bool(torch.equal(d1[0][k], d1[1][k])) for k in d1[0]})
    fe = [xf.net.backbone(a[:, :, :608, :608].contiguous(), True, True) for _ in range(2)]
    print("backbone608", [bool(torch.equal(x, y)) for x, y in zip(fe[0], fe[1])])
    d2 = xf.detectAndComputeDense(b, top_k=4096)
    m = [xf._batch_match_device(d1[0]["descriptors"], d2["descriptors"], -1) for _ in range(2)]
    nmv = m[0][2].tolist()      # idx buffers are torch.empty beyond n_matches: compare the valid prefixes only
    print("match", bool(torch.equal(m[0][2], m[1][2])) and all(bool(torch.equal(m[0][q][p, :nmv[p]], m[1][q][p, :nmv[p]])) for q in (0, 1) for p in range(len(nmv))), nmv)
    r = [xf._refine_device(d1[0], d2, m[0][0], m[0][1], m[0][2], 0.25) for _ in range(2)]
    n = r[0][1].tolist()
    print("refine n_out", n, r[1][1].tolist(), "rows equal", [bool(torch.equal(r[0][0][p, :n[p]], r[1][0][p, :n[p]])) for p in range(len(n))])
    # forced matches to exercise many rows
    N = d1[0]["keypoints"].shape[1]
    i0 = torch.arange(N, device="cuda")[None].repeat(8, 1); i1 = torch.stack([torch.randperm(N, device="cuda") for _ in range(8)])
    nm = torch.full((8,), N, dtype=torch.int32, device="cuda")
    r = [xf._refine_device(d1[0], d2, i0, i1, nm, 0.0) for _ in range(2)]
    print("forced refine equal", bool(torch.equal(r[0][0], r[1][0])), r[0][1].tolist())
    v = torch.cat([d1[0]["descriptors"][0], d2["descriptors"][0]], -1)
    f = [xf.net.fine_matcher(v) for _ in range(2)]
    print("fine_matcher equal", bool(torch.equal(f[0], f[1])))
