#!/usr/bin/env python3
"""Every matrix-core kernel of the step ALONE in a tight loop, cold-started (xfh_debug_cold_start), every result compared on the device with its first result:
the sensitive form of the cold-instruction-cache torture (tools/head_soak.py saw the split-bf16 key-point head fail at 1.5e-3 per launch this way, where whole
steps hide it).  With a library of `build.py --shift N` (XFH_LIB_PATH) the kernels sit 4 N bytes further along the instruction-cache lines: tools/shift_scan.sh kernels.
    python tools/conv_cold_scan.py [seconds per kernel]"""
import ctypes as C, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from accelerated_features_amd import XFeat, _lib
from accelerated_features_amd.spec import CONV_INDEX, CONV_BY_NAME
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
lib = _lib.load()
sd = fixtures.synthetic_state_dict(0)
xf = XFeat(weights=sd, top_k=4096)
h = xf.net.handle()
B, H, W = 64, 480, 640
P = lambda t: C.c_void_p(t.data_ptr())
g = torch.Generator(device="cuda").manual_seed(3)
DIV = {"block2.0": 4, "block3.0": 4, "block3.1": 8, "block_fusion.0": 8, "block4.0": 8, "block4.1": 16, "block5.0": 16, "block5.1": 32}
KERNEL = {"block2.0": "conv_bx_kernel<24,24,fx>", "block3.0": "conv_bxs2_kernel<24,fx>", "block3.1": "conv_bx64_kernel<64,0,fx> (1/8)", "block_fusion.0": "conv_bx64_kernel<64,0,fx>",
          "block4.0": "conv_bx64s2_kernel<1>", "block4.1": "conv_bx64_kernel<64,0,fx> (1/16)", "block5.0": "conv_bx64s2_kernel<2>", "block5.1": "conv_wino_kernel<128>"}
res = []
with torch.inference_mode():
    for name in DIV:
        c = CONV_BY_NAME[name]
        hin, win = H // DIV[name], W // DIV[name]
        x = torch.relu(torch.randn(B, c.cin, hin, win, device="cuda", generator=g))
        y = torch.empty(B, c.cout, (hin - 1) // c.stride + 1, (win - 1) // c.stride + 1, device="cuda")
        call = lambda: lib.xfh_conv_layer(h, CONV_INDEX[name], P(x), B, hin, win, P(y), 0, None)
        lib.xfh_debug_cold_start(0)
        assert call() == 0, lib.xfh_last_error()
        want = y.clone()
        lib.xfh_debug_cold_start(1)
        bad = torch.zeros(1, dtype=torch.int64, device="cuda")
        t0 = time.time(); n = 0
        while time.time() - t0 < secs:
            for _ in range(200):
                call()
                bad += (y != want).any()
                n += 1
            torch.cuda.synchronize()
        res.append((KERNEL[name], n, int(bad)))
    # the matcher's fp16 sweep + refine on the bench batch's descriptors
    lib.xfh_debug_cold_start(0)
    xb = torch.cat([fixtures.texture_images(8, H, W, seed=77)] * 8).cuda()
    kp, sc, de, nv, nc, cap, hw, d16 = xf._detect_device(xb, 4096, 0.05, want_f16=True)
    i0, i1, nm = xf.match_pairs_device(de, nv, -1, d16)
    w0, w1, wn = i0.clone(), i1.clone(), nm.clone()
    lo = int(nm.min())
    lib.xfh_debug_cold_start(1)
    bad = torch.zeros(1, dtype=torch.int64, device="cuda")
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(100):
            i0, i1, nm = xf.match_pairs_device(de, nv, -1, d16)
            bad += (i0[:, :lo] != w0[:, :lo]).any() | (i1[:, :lo] != w1[:, :lo]).any() | (nm != wn).any()
            n += 1
        torch.cuda.synchronize()
    res.append(("xfh_match_mnn (mnn_f16_sweep + refine; cold hook in --shift builds only)", n, int(bad)))
lib.xfh_debug_cold_start(0)
print(f"{os.path.basename(_lib.LIB_PATH)}: cold-started launches / launches with a wrong result: " + "; ".join(f"{k}: {n} / {b}" for k, n, b in res), flush=True)
sys.exit(1 if any(b for _, _, b in res) else 0)
