#!/usr/bin/env python3
"""Phase timeline of conv_bx_kernel (split-bf16 conv) from in-kernel s_memtime stamps (debug): per workgroup and tile
[start, staged, barrier passed, MFMAs done, stores issued, barrier passed]."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from accelerated_features_amd import XFeat, _lib
from accelerated_features_amd.spec import CONVS, CONV_INDEX
name = sys.argv[1] if len(sys.argv) > 1 else "block2.0"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
xf = XFeat(weights=fixtures.synthetic_state_dict(0)); lib = _lib.load(); h = xf.net.handle()
c = next(c for c in CONVS if c.name == name)
hin, win = 120, 160
x = torch.randn(B, c.cin, hin, win, device="cuda"); y = torch.empty(B, c.cout, hin, win, device="cuda")
tr = torch.zeros(64 * 4096, dtype=torch.int64, device="cuda")
def run():
    assert lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(x.data_ptr()), B, hin, win, C.c_void_p(y.data_ptr()), 10, None) == 0
for _ in range(3): run()
torch.cuda.synchronize()
lib.xfh_debug_trace(h, C.c_void_p(tr.data_ptr())); run(); torch.cuda.synchronize(); lib.xfh_debug_trace(h, None)
t = tr.cpu().numpy().reshape(-1, 64); t = t[t[:, 0] != 0]
print(name, "workgroups", len(t))
la = t[:, 62]; print("LDS_ALLOC values:", np.unique(la, return_counts=True)); print("first 32 workgroups: lds base", (la[:32] & 0xff))
t0 = t[:, 0].min()
st = t[:, :60].reshape(len(t), 10, 6).astype(np.float64)
ntile = (st[:, :, 5] != 0).sum(axis=1)
print("tiles per workgroup:", np.bincount(ntile.astype(int)))
print("kernel span (s_memtime ticks):", t[:, :60].max() - t0)
d = np.diff(st, axis=2)          # (wg, tile, 5)
valid = st[:, :, 5] != 0
for k, nm in enumerate(("stage (loads, split, LDS writes)", "barrier 1", "MFMA loop", "bias + stores", "barrier 2")):
    v = d[:, :, k][valid]
    print(f"{nm:34s} mean {v.mean():8.0f}  p10 {np.percentile(v, 10):8.0f}  p90 {np.percentile(v, 90):8.0f}")
per = (st[:, :, 5] - st[:, :, 0])[valid]
print("whole tile mean %.0f" % per.mean())
for k in range(10):
    v = valid[:, k]
    if v.any():
        print("tile", k, "phase means", d[v, k, :].mean(axis=0).round(0), "start", (st[v, k, 0] - t0).mean().round(0))
