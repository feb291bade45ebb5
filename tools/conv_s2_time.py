#!/usr/bin/env python3
"""block4.0 / block5.0 stand-alone: f32-MFMA direct kernel (xfh_conv_layer variant 0 with option bx = 5) against conv_bx64s2_kernel (variant 10), by batch."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from accelerated_features_amd import XFeat, _lib
from accelerated_features_amd.spec import CONVS, CONV_INDEX
xf = XFeat(weights=fixtures.synthetic_state_dict(0)); lib = _lib.load(); h = xf.net.handle()
xf.set_option("bx", 5)
for name, (hin, win) in (("block4.0", (60, 80)), ("block5.0", (30, 40))):
    c = next(c for c in CONVS if c.name == name)
    for B in (1, 2, 4, 8, 16, 32, 64):
        x = torch.randn(B, c.cin, hin, win, device="cuda"); y = torch.empty(B, c.cout, hin // 2, win // 2, device="cuda")
        res = []
        for variant in (0, 10):
            def run():
                assert lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(x.data_ptr()), B, hin, win, C.c_void_p(y.data_ptr()), variant, None) == 0
            for _ in range(5): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): run()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / 50)
        print(f"{name} B={B:3d}: f32 direct {res[0]:6.1f} us   split-bf16 s2 {res[1]:6.1f} us", flush=True)
