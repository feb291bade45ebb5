#!/usr/bin/env python3
"""Row timeline of conv_bx64s2_kernel (second unit of every workgroup) from in-kernel s_memtime stamps (debug).   [block4.0 | block5.0]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from accelerated_features_amd import XFeat, _lib
from accelerated_features_amd.spec import CONVS, CONV_INDEX
name = sys.argv[1] if len(sys.argv) > 1 else "block4.0"; B = 64
xf = XFeat(weights=fixtures.synthetic_state_dict(0)); lib = _lib.load(); h = xf.net.handle()
c = next(c for c in CONVS if c.name == name)
hin, win = (60, 80) if name == "block4.0" else (30, 40)
x = torch.randn(B, c.cin, hin, win, device="cuda"); y = torch.empty(B, c.cout, hin // 2, win // 2, device="cuda")
tr = torch.zeros(64 * 4096, dtype=torch.int64, device="cuda")
def run():
    assert lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(x.data_ptr()), B, hin, win, C.c_void_p(y.data_ptr()), 10, None) == 0
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(name, "stand-alone: %.1f us per launch" % (e0.elapsed_time(e1) * 1e3 / 20))
lib.xfh_debug_trace(h, C.c_void_p(tr.data_ptr())); run(); torch.cuda.synchronize(); lib.xfh_debug_trace(h, None)
t = tr.cpu().numpy().reshape(-1, 64).astype(np.float64); t = t[t[:, 0] != 0]
print("workgroups with a second unit", len(t))
rows = t[:, 1:49].reshape(len(t), 12, 4)
print("unit start -> first row start (zero acc, stage chunk 0): %.0f" % (rows[:, 0, 0] - t[:, 0]).mean())
print("per row means: vmcnt wait + barrier | DMA issue, fragment reads, MFMAs + staging | to next row start")
for r in range(12):
    nxt = rows[:, r + 1, 0] if r < 11 else t[:, 50]
    print(r, "%.0f %.0f %.0f" % ((rows[:, r, 1] - rows[:, r, 0]).mean(), (rows[:, r, 2] - rows[:, r, 1]).mean(), (nxt - rows[:, r, 2]).mean()))
print("whole unit (start -> stores issued): %.0f ; end barrier %.0f" % ((t[:, 50] - t[:, 0]).mean(), (t[:, 51] - t[:, 50])[t[:, 51] > 0].mean()))
