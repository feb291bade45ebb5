#!/usr/bin/env python3
"""Phase timeline of the MFMA conv kernel from in-kernel s_memtime stamps (debug)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from accelerated_features_amd import XFeat, _lib
from accelerated_features_amd.spec import CONVS, CONV_INDEX
name = sys.argv[1] if len(sys.argv) > 1 else "block3.1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
VAR = int(sys.argv[3]) if len(sys.argv) > 3 else 0
DIV = {"block2.0": 4, "block3.0": 4, "block3.1": 8, "block4.0": 8, "block4.1": 16, "block5.0": 16, "block5.1": 32, "block_fusion.0": 8, "block2.1": 4, "block4.2": 16, "block5.2": 32}
xf = XFeat(weights=fixtures.synthetic_state_dict(0)); lib = _lib.load(); h = xf.net.handle()
c = next(c for c in CONVS if c.name == name); d = DIV[name]
hin, win = 480 // d, 640 // d
hout, wout = (hin - 1) // c.stride + 1, (win - 1) // c.stride + 1
x = torch.randn(B, c.cin, hin, win, device="cuda"); y = torch.empty(B, c.cout, hout, wout, device="cuda")
tr = torch.zeros(24 * 40000, dtype=torch.int64, device="cuda")
def run():
    assert lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(x.data_ptr()), B, hin, win, C.c_void_p(y.data_ptr()), VAR, None) == 0
for _ in range(3): run()
torch.cuda.synchronize()
lib.xfh_debug_trace(h, C.c_void_p(tr.data_ptr())); run(); torch.cuda.synchronize(); lib.xfh_debug_trace(h, None)
t = tr.cpu().numpy().reshape(-1, 24); t = t[t[:, 0] != 0]
print(name, "workgroups", len(t))
t0 = t[:, 0].min()
FREQ = 100e6   # s_memtime ticks at 100 MHz on gfx9 (constant clock)
span = (t[:, 21].max() - t0)
print("kernel span ticks", span, "=> us at 100MHz:", span / 100)
start = (t[:, 0] - t0); end = (t[:, 21] - t0)
dur = end - start
nch = int((t[0, 2:19] != 0).sum())
print("chunks", nch)
pro = t[:, 2] - t[:, 0]; main = t[:, 20] - t[:, 2]; epi = t[:, 21] - t[:, 20]
per = np.diff(t[:, 2:2 + nch], axis=1)
print("per-WG duration: mean %.1f  min %.1f  max %.1f ticks" % (dur.mean(), dur.min(), dur.max()))
print("prologue (to first barrier passed) mean %.1f max %.1f | main mean %.1f | epilogue mean %.1f max %.1f" % (pro.mean(), pro.max(), main.mean(), epi.mean(), epi.max()))
print("per-chunk mean", per.mean(axis=0).round(1))
if VAR >= 2:
    d = t[:, [20, 10, 11, 12, 13, 21]].astype(np.float64)
    print("epilogue (wave 0): wait for all waves %.0f | T + X writes %.0f | barrier %.0f | X reads + sums %.0f | stores %.0f" % tuple(np.diff(d, axis=1).mean(axis=0)))
order = np.argsort(start)
print("start times (sorted, every 64th):", start[order][::64][:30])
print("end   times (sorted, every 64th):", np.sort(end)[::64][:30])
rt = (t[:, 23] - t[:, 19]).astype(np.float64); st = (t[:, 21] - t[:, 0]).astype(np.float64)
print("shader clock estimate (s_memtime / s_memrealtime@100MHz): median %.3f GHz" % (np.median(st / rt) * 0.1))
xcc = t[:, 22] >> 32; hw = t[:, 22] & 0xffffffff
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7
key = xcc * 1000 + se * 16 + cu
u, cnt = np.unique(key, return_counts=True)
print("distinct (xcc,se,cu):", len(u), "WGs per CU: min %d max %d" % (cnt.min(), cnt.max()), np.bincount(cnt))
