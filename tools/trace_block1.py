#!/usr/bin/env python3
"""Stage timeline of block1_bx_kernel (second tile of every workgroup) from in-kernel s_memtime stamps (debug)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from accelerated_features_amd import XFeat, _lib
xf = XFeat(weights=fixtures.synthetic_state_dict(0)); lib = _lib.load(); h = xf.net.handle()
x = fixtures.texture_images(8, 480, 640, seed=3); x = torch.cat([x] * 8).cuda()
tr = torch.zeros((1 << 21) + 16 * 4096, dtype=torch.int64, device="cuda")      # block1 stamps start 2 Mi entries in
for _ in range(3): xf.net(x)
torch.cuda.synchronize()
lib.xfh_debug_trace(h, C.c_void_p(tr.data_ptr())); xf.net(x); torch.cuda.synchronize(); lib.xfh_debug_trace(h, None)
t = tr[(1 << 21):].cpu().numpy().reshape(-1, 16).astype(np.float64); t = t[(t[:, 0] != 0) & (t[:, 6] != 0)]
print("workgroups", len(t))
d = np.diff(t[:, :7], axis=1)
for k, nm in enumerate(("stage 0 gray", "stage 1 conv1 (VALU) + skip avg", "stage 2 conv2 (VALU) + split", "stage 3 conv3", "stage 4 conv4 (MFMA) + stores", "end")):
    print(f"{nm:34s} mean {d[:, k].mean():8.0f}  p10 {np.percentile(d[:, k], 10):8.0f}  p90 {np.percentile(d[:, k], 90):8.0f}")
print("tile mean %.0f" % (t[:, 6] - t[:, 0]).mean())

