#!/usr/bin/env python3
"""Winograd conv path vs generic kernel (accuracy) and vs the direct MFMA kernel (time), per layer."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from accelerated_features_amd import XFeat, _lib
from accelerated_features_amd.spec import CONVS, CONV_INDEX
sys.path.insert(0, os.path.join(ROOT, "tools"))
from microbench import DIV, time_fn

xf = XFeat(weights=fixtures.synthetic_state_dict(), top_k=4096)
lib = _lib.load(); h = xf.net.handle()
names = sys.argv[1:] or ["block2.0", "block3.1", "block4.1", "block5.1"]
for (B, H, W) in ((64, 480, 640), (8, 1312, 1312), (3, 480, 640)):
    for name in names:
        c = next(c for c in CONVS if c.name == name)
        d = DIV[name]; hin, win = H // d, W // d
        x = torch.randn(B, c.cin, hin, win, device="cuda")
        ref = torch.empty(B, c.cout, hin, win, device="cuda"); y = torch.full_like(ref, float("nan"))
        assert lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(x.data_ptr()), B, hin, win, C.c_void_p(ref.data_ptr()), 1, None) == 0
        rc = lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(x.data_ptr()), B, hin, win, C.c_void_p(y.data_ptr()), 2, None)
        if rc: print(name, "rc", rc, lib.xfh_last_error()); continue
        torch.cuda.synchronize()
        err = float((y - ref).abs().nan_to_num(1e9).max())
        t = {}
        errs = {2: err}
        print('note: variant 0 is the Winograd path unless the handle option wino is 0', end=' ')
        for v in (0, 2, 3, 4, 5, 6):
            if v > 2:
                y.fill_(float("nan"))
                if lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(x.data_ptr()), B, hin, win, C.c_void_p(y.data_ptr()), v, None):
                    continue
                errs[v] = float((y - ref).abs().nan_to_num(1e9).max())
            t[v] = time_fn(lambda: lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(x.data_ptr()), B, hin, win, C.c_void_p(y.data_ptr()), v, None))
        print(f"{name:15s} B={B:2d} {hin:4d}x{win:<4d} direct {t[0]:7.1f} us | " + " | ".join(f"cfg{v-2} {t[v]:6.1f} us err {errs[v]:.1e}" for v in t if v >= 2), flush=True)
