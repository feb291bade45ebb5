#!/usr/bin/env python3
"""Time single kernels of the hot path on the GPU (events on the launch stream).

    python tools/microbench.py conv block3.1 --batch 64            # one conv layer at VGA scale
    python tools/microbench.py match --pairs 32 --n 4096
    python tools/microbench.py all                                   # every MFMA conv layer + match
"""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import fixtures  # noqa: E402
from accelerated_features_amd import XFeat, _lib  # noqa: E402
from accelerated_features_amd.spec import CONVS, CONV_INDEX  # noqa: E402

# input resolution divisor of every conv layer relative to the image
DIV = {"block1.0": 1, "block1.1": 1, "block1.2": 2, "block1.3": 2, "block2.0": 4, "block2.1": 4, "block3.0": 4,
       "block3.1": 8, "block3.2": 8, "block4.0": 8, "block4.1": 16, "block4.2": 16, "block5.0": 16, "block5.1": 32,
       "block5.2": 32, "block5.3": 32, "block_fusion.0": 8, "block_fusion.1": 8, "block_fusion.2": 8}


def time_fn(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3   # us


def bench_conv(xf, lib, name, B, H, W, variant=0, iters=20):
    c = next(c for c in CONVS if c.name == name)
    d = DIV[name]
    hin, win = H // d, W // d
    hout, wout = (hin - 1) // c.stride + 1, (win - 1) // c.stride + 1
    x = torch.randn(B, c.cin, hin, win, device="cuda")
    y = torch.empty(B, c.cout, hout, wout, device="cuda")
    h = xf.net.handle()

    def fn():
        rc = lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(x.data_ptr()), B, hin, win, C.c_void_p(y.data_ptr()), variant, None)
        assert rc == 0, lib.xfh_last_error()
    us = time_fn(fn, iters)
    fl = 2.0 * B * hout * wout * c.cout * c.cin * c.k * c.k
    by = 4.0 * (x.numel() + y.numel())
    print(f"{name:16s} {c.cin:3d}->{c.cout:3d} k{c.k} s{c.stride} out {hout}x{wout} B={B}: {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  "
          f"{by / us / 1e3:7.1f} GB/s(act)")
    return us


def bench_match(xf, P, N, iters=10):
    d = torch.nn.functional.normalize(torch.randn(2 * P, N, 64, device="cuda"), dim=-1)
    nv = torch.full((2 * P,), N, dtype=torch.int32, device="cuda")
    us = time_fn(lambda: xf.match_pairs_device(d, nv, -1), iters)
    print(f"match P={P} N={N}: {us:9.1f} us  {2.0 * P * N * N * 64 / us / 1e6:7.1f} TFLOP/s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what")
    ap.add_argument("layer", nargs="?")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--H", type=int, default=480)
    ap.add_argument("--W", type=int, default=640)
    ap.add_argument("--pairs", type=int, default=32)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    xf = XFeat(weights=fixtures.synthetic_state_dict(0))
    lib = _lib.load()
    if a.what == "conv":
        bench_conv(xf, lib, a.layer, a.batch, a.H, a.W, 0, a.iters)
    elif a.what == "match":
        bench_match(xf, a.pairs, a.n, a.iters)
    elif a.what == "all":
        tot = 0
        for name in DIV:
            tot += bench_conv(xf, lib, name, a.batch, a.H, a.W, 0, a.iters)
        print(f"sum of conv layers: {tot:.1f} us")
        bench_match(xf, a.pairs, a.n)


if __name__ == "__main__":
    main()
