#!/usr/bin/env python3
"""Run-to-run bit determinism of the LighterGlue matcher (40 repeats at three sizes / pruning thresholds)."""
import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fixtures
from accelerated_features_amd.lighterglue import LighterGlue
lg = LighterGlue(weights=fixtures.lighterglue_state_dict(0))
bad = 0
for (n0, n1, prune) in ((4096, 4096, 1536), (1500, 2100, -1), (700, 640, 200)):
    k0, d0, s0, k1, d1, s1 = fixtures.lighterglue_inputs(n0, n1, seed=n0)
    args = (k0.cuda(), d0.cuda(), s0.tolist(), k1.cuda(), d1.cuda(), s1.tolist(), 0.01, prune)
    m, s, c = lg.match_device(*args); n = int(c.item()); rm, rs = m[:n].clone(), s[:n].clone()
    for it in range(40):
        m, s, c = lg.match_device(*args)
        if int(c.item()) != n or not torch.equal(m[:n], rm) or not torch.equal(s[:n], rs):
            bad += 1
    print(n0, n1, prune, "matches", n, "mismatching repeats", bad)
print("LG determinism:", "OK" if bad == 0 else "FAILED")
