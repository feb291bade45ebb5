#!/usr/bin/env python3
"""Time of the split-bf16 conv (xfh_conv_layer variant 10) at the bench shape, 50 back-to-back launches (XFH_BX_LAG experiments)."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import fixtures
from accelerated_features_amd import XFeat, _lib
from accelerated_features_amd.spec import CONVS, CONV_INDEX
from microbench import time_fn
xf = XFeat(weights=fixtures.synthetic_state_dict(), top_k=4096)
lib = _lib.load(); h = xf.net.handle()
name = "block2.0"; c = next(c for c in CONVS if c.name == name)
for (B, hin, win) in ((64, 120, 160),):
    x = torch.randn(B, c.cin, hin, win, device="cuda"); y = torch.empty(B, c.cout, hin, win, device="cuda")
    t = time_fn(lambda: lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(x.data_ptr()), B, hin, win, C.c_void_p(y.data_ptr()), 10, None), iters=50)
    print(f"{name} B={B} {hin}x{win}: {t:7.1f} us (XFH_BX_LAG={os.environ.get('XFH_BX_LAG', 'default')})", flush=True)
