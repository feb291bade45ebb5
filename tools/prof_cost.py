import os, sys, time
import torch
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures, bench
from accelerated_features_amd import XFeat, _lib
from accelerated_features_amd.streaming import FrameStream
lib = _lib.load()
B = 64
xfs = [XFeat(weights=fixtures.synthetic_state_dict(0), top_k=4096) for _ in range(2)]
x = bench.make_frames(B, seed=1000).cuda()
fs = FrameStream(xfeats=xfs, top_k=4096)
hs = [xf.net.handle() for xf in xfs]
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        if fs.in_flight == fs.lanes: fs.result()
        fs.submit(x)
    fs.drain(); torch.cuda.synchronize()
    return B * n / (time.perf_counter() - t0)
run(150)
for rnd in range(3):
    a = run(20)
    for h in hs: lib.xfh_profile_select(h, _lib.PROF_BLOCK1)
    b = run(20)
    b2 = run(20)
    for h in hs: lib.xfh_profile_select(h, _lib.PROF_NONE)
    print(f"round {rnd}: no profiler {a:.0f}   block1 events (first window: events created) {b:.0f}   (second) {b2:.0f}", flush=True)
