#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do timeout 900 python -m pytest tests -m gpu -q --tb=line -p no:cacheprovider 2>&1 | tail -2; done
python - <<'P'
import sys, os, torch, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import fixtures
from accelerated_features_amd import XFeat
xf = XFeat(weights=fixtures.synthetic_state_dict(0), top_k=4096, detection_threshold=0.05)
x = torch.cat([fixtures.texture_images(8, 480, 640, seed=s) for s in range(8)]).cuda()
ref = None; bad = 0
for rep in range(200):
    feats, logits, rel = xf.net(x)
    cur = (feats.clone(), logits.clone(), rel.clone())
    if ref is None: ref = cur
    else:
        for a, b in zip(ref, cur):
            if not torch.equal(a, b): bad += 1
print("200 backbone repetitions at B=64 VGA: mismatching tensors:", bad)
P
