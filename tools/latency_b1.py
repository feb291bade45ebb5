#!/usr/bin/env python3
"""Single-pair latency (realtime_demo.py use case): detectAndCompute on 1 or 2 VGA frames + match, wall clock per call."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from accelerated_features_amd import XFeat
xf = XFeat(weights=fixtures.synthetic_state_dict(), top_k=4096)
x = fixtures.texture_images(2, 480, 640, seed=3).cuda()
def bench(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("detectAndCompute B=1          : %.3f ms" % bench(lambda: xf.detectAndCompute(x[:1], top_k=4096)))
print("detectAndCompute B=2          : %.3f ms" % bench(lambda: xf.detectAndCompute(x, top_k=4096)))
o = xf.detectAndCompute(x, top_k=4096)
print("match 4096x4096               : %.3f ms" % bench(lambda: xf.match(o[0]['descriptors'], o[1]['descriptors'], 0.82)))
print("match_xfeat (2 frames + match): %.3f ms" % bench(lambda: xf.match_xfeat(x[0:1], x[1:2], top_k=4096)))
def dev_only():
    kp, sc, de, nv, nc, cap, hw = xf._detect_device(x, 4096, 0.05)
    xf.match_pairs_device(de, nv, -1)
print("device-only B=2 detect + match (no read-back): %.3f ms" % bench(dev_only))
from accelerated_features_amd.graphs import CapturedSparsePipeline
for B in (1, 2):
    pipe = CapturedSparsePipeline(xf, batch=B, height=480, width=640, top_k=4096, match=(B == 2))
    print("hipGraph B=%d detect%s incl. read-back: %.3f ms" % (B, " + match" if B == 2 else "", bench(lambda: pipe(x[:B]))))
# equality with the eager path
pipe = CapturedSparsePipeline(xf, batch=2, height=480, width=640, top_k=4096, match=True)
o = pipe(x)
e = xf.detectAndCompute(x, top_k=4096)
for b in range(2):
    n = o['n_valid'][b]
    assert n == e[b]['keypoints'].shape[0] and torch.equal(o['keypoints'][b, :n], e[b]['keypoints']) and torch.equal(o['descriptors'][b, :n], e[b]['descriptors'])
i0, i1 = xf.match(e[0]['descriptors'], e[1]['descriptors'], min_cossim=-1)
assert torch.equal(o['matches'][0][0], i0) and torch.equal(o['matches'][0][1], i1)
print("graph replay == eager path: OK")
