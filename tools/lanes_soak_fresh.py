#!/usr/bin/env python3
"""The start-up of the two-lane path: fresh FrameStream (fresh handles, workspaces, streams) again and again, 40 batches each, every result against the synchronous one."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from accelerated_features_amd import XFeat
from accelerated_features_amd.streaming import FrameStream
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
LANES = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sd = fixtures.synthetic_state_dict(0)
ref = XFeat(weights=sd, top_k=4096)
x = torch.cat([fixtures.texture_images(8, 480, 640, seed=77)] * 8).cuda()
kp0, sc0, de0, nv0, nc0, cap, hw, d16 = ref._detect_device(x, 4096, 0.05, want_f16=True)
i00, i10, nm0 = ref.match_pairs_device(de0, nv0, -1, d16)
nv_h, nm_h, nc_h = nv0.cpu(), nm0.cpu(), nc0.cpu()
nbad = 0
for rnd in range(rounds):
    fs = FrameStream(weights=sd, top_k=4096, lanes=LANES)
    got = []
    for step in range(40):
        if fs.in_flight == fs.lanes: got.append(fs.result())
        fs.submit(x)
    got += fs.drain()
    for step, r in enumerate(got):
        what = []
        if not torch.equal(r["n_valid"], nv_h): what.append("n_valid")
        if not torch.equal(r["n_candidates"], nc_h): what.append("n_candidates")
        if not torch.equal(r["n_matches"], nm_h): what.append(f"n_matches {(r['n_matches'] != nm_h).nonzero().flatten().tolist()}")
        if not torch.equal(r["keypoints"], kp0): what.append(f"keypoints imgs {sorted(set((r['keypoints'] != kp0).flatten(1).any(1).nonzero().flatten().tolist()))}")
        if not torch.equal(r["scores"], sc0): what.append("scores")
        if not torch.equal(r["descriptors"], de0): what.append(f"descriptors imgs {sorted(set((r['descriptors'] != de0).flatten(1).any(1).nonzero().flatten().tolist()))}")
        for p in range(32):
            n = int(nm_h[p])
            if not (torch.equal(r["idx0"][p, :n], i00[p, :n]) and torch.equal(r["idx1"][p, :n], i10[p, :n])):
                what.append(f"idx pair {p}"); break
        if what:
            nbad += 1
            print(f"round {rnd} ticket {step}: " + "; ".join(what), flush=True)
    del fs
print(f"{rounds} fresh two-lane streams x 40 batches: {nbad} wrong results")
