#!/usr/bin/env python3
"""Per-layer GPU time of the MFMA convolutions INSIDE the real step (HIP events around each launch):
    python tools/layer_insitu.py            # all layers, B=64 VGA"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from accelerated_features_amd import XFeat, _lib
from accelerated_features_amd.spec import CONVS, CONV_INDEX
xf = XFeat(weights=fixtures.synthetic_state_dict(), top_k=4096)
lib = _lib.load(); h = xf.net.handle()
x = torch.cat([fixtures.texture_images(8, 480, 640, seed=5)] * 8).cuda()
names = ["block2.0", "block2.1", "block3.0", "block3.1", "block4.0", "block4.1", "block4.2", "block5.0", "block5.1", "block5.2", "block_fusion.0", "block_fusion.1"]
for _ in range(3): xf._detect_device(x, 4096, 0.05)
tot = 0.0
for n in names:
    lib.xfh_profile_select(h, 100 + CONV_INDEX[n])
    for _ in range(5): xf._detect_device(x, 4096, 0.05)
    torch.cuda.synchronize()
    nl, ms, fl, by = C.c_int(), C.c_double(), C.c_double(), C.c_double()
    lib.xfh_profile_read(h, C.byref(nl), C.byref(ms), C.byref(fl), C.byref(by))
    us = 1e3 * ms.value / max(nl.value, 1); tot += us
    print(f"{n:16s} {us:8.1f} us   ({nl.value} launches)")
lib.xfh_profile_select(h, 0)
print(f"sum {tot:.1f} us")
