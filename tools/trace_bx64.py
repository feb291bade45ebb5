#!/usr/bin/env python3
"""Row timeline of conv_bx64_kernel (second tile of every workgroup) from in-kernel s_memtime stamps (debug)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from accelerated_features_amd import XFeat, _lib
from accelerated_features_amd.spec import CONVS, CONV_INDEX
name = "block_fusion.0"; B = 64
xf = XFeat(weights=fixtures.synthetic_state_dict(0)); lib = _lib.load(); h = xf.net.handle()
c = next(c for c in CONVS if c.name == name)
hin, win = 60, 80
x = torch.randn(B, c.cin, hin, win, device="cuda"); y = torch.empty(B, c.cout, hin, win, device="cuda")
tr = torch.zeros(64 * 4096, dtype=torch.int64, device="cuda")
def run():
    assert lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(x.data_ptr()), B, hin, win, C.c_void_p(y.data_ptr()), 10, None) == 0
for _ in range(3): run()
torch.cuda.synchronize()
lib.xfh_debug_trace(h, C.c_void_p(tr.data_ptr())); run(); torch.cuda.synchronize(); lib.xfh_debug_trace(h, None)
t = tr.cpu().numpy().reshape(-1, 64).astype(np.float64); t = t[t[:, 0] != 0]
print(name, "workgroups with a second tile", len(t))
rows = t[:, 1:49].reshape(len(t), 12, 4)
ok = rows[:, :, 3].min(axis=1) > 0
rows = rows[ok]; t = t[ok]
print("tile start -> first row start (zero acc, stage chunk 0): %.0f" % (rows[:, 0, 0] - t[:, 0]).mean())
print("per row means: wait+barrier | issue DMA/loads | MFMAs issued | to next row start (stage at chunk ends)")
for r in range(12):
    nxt = rows[:, r + 1, 0] if r < 11 else t[:, 50]
    print(r, "%.0f %.0f %.0f %.0f" % ((rows[:, r, 1] - rows[:, r, 0]).mean(), (rows[:, r, 2] - rows[:, r, 1]).mean(), (rows[:, r, 3] - rows[:, r, 2]).mean(), (nxt - rows[:, r, 3]).mean()))
print("whole tile (start -> stores issued): %.0f ; end barrier %.0f" % ((t[:, 50] - t[:, 0]).mean(), (t[:, 51] - t[:, 50])[t[:, 51] > 0].mean()))
