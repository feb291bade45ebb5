#!/usr/bin/env python3
"""Layer timeline of head_bx_kernel<true> (wave 0, second tile of every workgroup) from in-kernel s_memtime stamps (debug)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from accelerated_features_amd import XFeat, _lib
xf = XFeat(weights=fixtures.synthetic_state_dict(0)); lib = _lib.load(); h = xf.net.handle()
x = fixtures.texture_images(8, 480, 640, seed=3); x = torch.cat([x] * 8).cuda()
OFF = (1 << 21) + (1 << 16)
tr = torch.zeros(OFF + 16 * 4096, dtype=torch.int64, device="cuda")
for _ in range(3): xf.net(x)
torch.cuda.synchronize()
lib.xfh_debug_trace(h, C.c_void_p(tr.data_ptr())); xf.net(x); torch.cuda.synchronize(); lib.xfh_debug_trace(h, None)
t = tr[OFF:].cpu().numpy().reshape(-1, 16).astype(np.float64); t = t[(t[:, 0] != 0) & (t[:, 5] != 0)]
print("workgroups", len(t))
d = np.diff(t[:, :6], axis=1)
for k, nm in enumerate(("layer 1 (input from registers)", "layer 2", "layer 3", "layer 4 (96 outputs)", "softmax + heat stores")):
    print(f"{nm:34s} mean {d[:, k].mean():8.0f}  p10 {np.percentile(d[:, k], 10):8.0f}  p90 {np.percentile(d[:, k], 90):8.0f}")
print("wave-tile mean %.0f" % (t[:, 5] - t[:, 0]).mean())
