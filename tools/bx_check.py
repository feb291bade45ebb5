#!/usr/bin/env python3
"""Split-bf16 conv (xfh_conv_layer variant 10) against an fp64 convolution of the same folded weights, next to the generic fp32
kernel (variant 1) and the Winograd f32-MFMA kernel (variant 2); then timing at the bench shapes."""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import fixtures
from accelerated_features_amd import XFeat, _lib
from accelerated_features_amd.spec import CONVS, CONV_INDEX, BN_EPS
from microbench import DIV, time_fn

sd = fixtures.synthetic_state_dict()
xf = XFeat(weights=sd, top_k=4096)
lib = _lib.load(); h = xf.net.handle()
names = sys.argv[1:] or ["block2.0", "block2.1"]
g = torch.Generator(device="cuda").manual_seed(5)


def truth64(name, x):
    c = next(c for c in CONVS if c.name == name)
    w = sd[f"{name}.layer.0.weight"].double().cuda()
    rm = sd[f"{name}.layer.1.running_mean"].double().cuda(); rv = sd[f"{name}.layer.1.running_var"].double().cuda()
    sc = 1.0 / torch.sqrt(rv + BN_EPS)
    wf = (w * sc[:, None, None, None]).float().double()          # the library folds in fp64 and rounds the weights to fp32
    bf = (-rm * sc).float().double()
    return torch.relu(torch.nn.functional.conv2d(x.double(), wf, bf, stride=c.stride, padding=1))


def run(name, x, variant):
    c = next(c for c in CONVS if c.name == name)
    B, _, hh, ww = x.shape
    y = torch.full((B, c.cout, (hh - 1) // c.stride + 1, (ww - 1) // c.stride + 1), float("nan"), device="cuda")
    rc = lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(x.data_ptr()), B, hh, ww, C.c_void_p(y.data_ptr()), variant, None)
    assert rc == 0, (name, variant, lib.xfh_last_error())
    torch.cuda.synchronize()
    return y


for name in names:
    c = next(c for c in CONVS if c.name == name)
    for (B, hh, ww) in ((2, 24, 32), (3, 41, 41), (1, 6, 10), (9, 30, 40), (8, 120, 160), (2, 7, 70)):
        for scale in (1.0, 30.0):
            x = torch.randn(B, c.cin, hh, ww, device="cuda", generator=g) * scale
            t = truth64(name, x)
            ref = float(t.abs().max())
            vs = (1, 2, 10) if c.stride == 1 else (1, 10)
            errs = {v: float((run(name, x, v).double() - t).abs().nan_to_num(1e9).max()) / ref for v in vs}
            print(f"{name} B={B} {hh}x{ww} scale {scale:4.0f}: max|err|/max|y|  generic {errs[1]:.2e}  " + (f"winograd {errs[2]:.2e}  " if 2 in errs else "") + f"split-bf16 {errs[10]:.2e}", flush=True)
for (B, H, W) in ((64, 480, 640), (8, 1312, 1312)):
    for name in names:
        c = next(c for c in CONVS if c.name == name)
        d = DIV[name]; hin, win = H // d, W // d
        x = torch.randn(B, c.cin, hin, win, device="cuda", generator=g)
        y = torch.empty(B, c.cout, (hin - 1) // c.stride + 1, (win - 1) // c.stride + 1, device="cuda")
        # variant 0 = the production dispatch (xf.set_option('bx', 0) times the f32-MFMA kernel of a stride-2 layer there)
        t = {v: time_fn(lambda: lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(x.data_ptr()), B, hin, win, C.c_void_p(y.data_ptr()), v, None)) for v in ((2, 10) if c.stride == 1 else (0, 10))}
        print(f"{name} B={B} {hin}x{win}: " + (f"winograd {t[2]:7.1f} us" if 2 in t else f"dispatch {t[0]:7.1f} us") + f"   split-bf16 {t[10]:7.1f} us", flush=True)
