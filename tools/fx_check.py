#!/usr/bin/env python3
"""The fp16-pair arithmetic (xfh_conv_layer variant 11, option fx) against an fp64 convolution, next to the generic fp32 kernel (1) and the
three-way bf16 split (10); timing at the bench shape; the whole backbone with option fx against the default one."""
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
names = [a for a in sys.argv[1:] if not a.startswith("-")] or ["block2.0", "block3.0", "block3.1", "block_fusion.0"]
g = torch.Generator(device="cuda").manual_seed(5)


def truth64(name, x):
    c = next(c for c in CONVS if c.name == name)
    w = sd[f"{name}.layer.0.weight"].double().cuda()
    rm = sd[f"{name}.layer.1.running_mean"].double().cuda(); rv = sd[f"{name}.layer.1.running_var"].double().cuda()
    sc = 1.0 / torch.sqrt(rv + BN_EPS)
    wf = (w * sc[:, None, None, None]).float().double()
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


VARS = {1: "generic fp32", 10: "bf16 x3", 11: "fp16 pair"}
for name in names:
    c = next(c for c in CONVS if c.name == name)
    for (B, hh, ww) in ((2, 24, 32), (3, 41, 44), (1, 6, 12), (9, 30, 40)):
        for scale, kind in ((1.0, "relu-normal"), (30.0, "relu-normal"), (1e-3, "relu-normal"), (300.0, "signed")):
            x = torch.randn(B, c.cin, hh, ww, device="cuda", generator=g) * scale
            if kind == "relu-normal": x = torch.relu(x)
            t = truth64(name, x)
            ref = float(t.abs().max())
            errs = {v: float((run(name, x, v).double() - t).abs().nan_to_num(1e9).max()) / ref for v in VARS}
            print(f"{name} B={B} {hh}x{ww} {kind} x {scale:g}: max|err|/max|y|  " + "  ".join(f"{VARS[v]} {errs[v]:.2e}" for v in VARS), flush=True)
for name in names:
    c = next(c for c in CONVS if c.name == name)
    B, H, W = 64, 480, 640
    d = DIV[name]; hin, win = H // d, W // d
    x = torch.relu(torch.randn(B, c.cin, hin, win, device="cuda", generator=g))
    y = torch.empty(B, c.cout, (hin - 1) // c.stride + 1, (win - 1) // c.stride + 1, device="cuda")
    for rnd in range(3):
        t = {v: time_fn(lambda: lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(x.data_ptr()), B, hin, win, C.c_void_p(y.data_ptr()), v, None), iters=30) for v in (10, 11)}
        print(f"{name} B={B} {hin}x{win}: bf16 x3 {t[10]:7.1f} us   fp16 pair {t[11]:7.1f} us", flush=True)
# whole backbone: option fx on a second model against the default
x = torch.cat([fixtures.texture_images(8, 480, 640, seed=77)] * 8).cuda()
a = XFeat(weights=sd, top_k=4096); b = XFeat(weights=sd, top_k=4096); b.set_option("fx", 15)
with torch.inference_mode():
    fa = a.net.backbone(x, want_logits=False, want_heat=True, want_invnorm=True)
    fb = b.net.backbone(x, want_logits=False, want_heat=True, want_invnorm=True)
for n, ta, tb in zip(("feats", "logits", "heat", "rel", "inv"), fa, fb):
    if ta is None: continue
    print(f"backbone fx vs default: {n}: max |diff| {float((ta - tb).abs().max()):.3e} at magnitude {float(ta.abs().max()):.3g}", flush=True)
ra = a.detectAndCompute(x[:8], top_k=4096); rb = b.detectAndCompute(x[:8], top_k=4096)
for i in range(8):
    ka, kb = ra[i]['keypoints'], rb[i]['keypoints']
    same = ka.shape == kb.shape and bool(torch.equal(ka, kb))
    print(f"image {i}: {ka.shape[0]} / {kb.shape[0]} key-points, identical lists: {same}" + ("" if same else f"; common {len(set(map(tuple, ka.tolist())) & set(map(tuple, kb.tolist())))}"), flush=True)
