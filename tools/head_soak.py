#!/usr/bin/env python3
"""Soak of the key-point head ALONE next to a foreign load on a second HIP stream, and the cold-start code-position scan.

Every launch's heat map (and logits) is compared on the device with the result of a quiet run; differing float4s are recorded
(iteration, index, bits) and decoded here into (image, cell, workgroup tile, wave, lane) so that a rare wrong block can be placed.
Variants (xfh_debug_head_soak): 0 = the split-bf16 head, 100 / 101 = the f32-MFMA heads (activation tile in LDS / register input = the default), 102 / 103 / 104 = the split head in the
fp16-pair arithmetic (three weight fragments | two | three and the pixel-side fragments through LDS);
1000 + s, 2000 + s, 3000 + s, 4000 + s, 5000 + s = 0, 100, 101, 102, 104 started on an invalidated instruction cache with the code moved by 4 s bytes (s = 0 .. 15).

    python tools/head_soak.py --variants 0,101 --foreign backbone,copy,none --max-seconds 45           # two streams
    python -m accelerated_features_amd.build --scan          # (the 3 x 16 scan instantiations live in libxfeat_hip_scan.so only)
    XFH_LIB_PATH=accelerated_features_amd/libxfeat_hip_scan.so python tools/head_soak.py --variants $(seq -s, 1000 1015),$(seq -s, 3000 3015) --foreign none --max-seconds 6 --logits 0

(The experiment builds behind profiles/r04_head_hazard/t1..t9 -- variants 1..26: reloads, pads, dumps, a dry first pass -- were removed after commit 9607d16.)
"""
import argparse
import ctypes as C
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures  # noqa: E402
from accelerated_features_amd import XFeat, _lib  # noqa: E402
from accelerated_features_amd.spec import CONV_INDEX, CONV_BY_NAME  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="0")
ap.add_argument("--foreign", default="backbone")
ap.add_argument("--iters", type=int, default=100000)
ap.add_argument("--chunk", type=int, default=250)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--logits", type=int, default=1)
ap.add_argument("--max-seconds", type=float, default=120.0, help="per configuration")
args = ap.parse_args()

B, H, W = args.batch, 480, 640
CAP = 1 << 16
lib = _lib.load()
sd = fixtures.synthetic_state_dict(0)
x = torch.cat([fixtures.texture_images(8, H, W, seed=77)] * (B // 8)).cuda()
xf = XFeat(weights=sd, top_k=4096)
h = xf.net.handle()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
gray = torch.empty(B, H, W, device="cuda")
coef = torch.empty(B, 2, device="cuda")
part = torch.empty(B * 128, dtype=torch.float64, device="cuda")
ncell = B * (H // 8) * (W // 8)


def run(variant, stream, iters, iter0, heat, heat_ref, logits, logits_ref, rep_h, rep_l, img=None):
    rc = lib.xfh_debug_head_soak(h, P(img), B, 3, H, W, P(gray), P(coef), P(part), P(heat), P(heat_ref), P(logits), P(logits_ref),
                                 variant, iters, iter0, P(rep_h), P(rep_l), CAP, C.c_void_p(stream.cuda_stream))
    assert rc == 0, lib.xfh_last_error()


# ---- foreign loads (second stream, own thread; ctypes / torch release the GIL while they launch) ----------------------------------
def foreign_thread(kind, stop):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st), torch.inference_mode():
        if kind == "backbone":
            other = XFeat(weights=sd, top_k=4096)
            fn = lambda: other.net.backbone(x, want_logits=False, want_heat=True, want_invnorm=True)
        elif kind == "backbone_f32heads":
            other = XFeat(weights=sd, top_k=4096); other.set_option("heads_f32", 1)
            fn = lambda: other.net.backbone(x, want_logits=False, want_heat=True, want_invnorm=True)
        elif kind == "copy":
            a = torch.empty(64 << 20, device="cuda"); b = torch.empty_like(a)
            fn = lambda: b.copy_(a)
        elif kind == "mm":
            a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda"); c = torch.empty_like(a)
            fn = lambda: torch.mm(a, b, out=c)
        elif kind == "valu":
            a = torch.rand(16 << 20, device="cuda")
            fn = lambda: torch.sin_(a)
        elif kind.startswith("conv:"):      # one layer of the network through xfh_conv_layer, e.g. conv:block4.1 (Winograd f32 MFMA), conv:block_fusion.0 (split bf16)
            name = kind[5:]
            c = CONV_BY_NAME[name]
            div = {"block2.0": 4, "block2.1": 4, "block3.0": 4, "block3.1": 8, "block4.0": 8, "block4.1": 16, "block4.2": 16, "block5.0": 16, "block5.1": 32,
                   "block_fusion.0": 8}[name]
            hin, win = H // div, W // div
            xi = torch.randn(B, c.cin, hin, win, device="cuda")
            yo = torch.empty(B, c.cout, (hin - 1) // c.stride + 1, (win - 1) // c.stride + 1, device="cuda")
            oh = XFeat(weights=sd, top_k=4096).net
            hh = oh.handle()

            def fn():
                rc = lib.xfh_conv_layer(hh, CONV_INDEX[name], P(xi), B, hin, win, P(yo), 0, C.c_void_p(st.cuda_stream))
                assert rc == 0, lib.xfh_last_error()
        else:
            raise SystemExit(f"unknown foreign load {kind}")
        evs = []
        while not stop.is_set():
            fn()
            ev = torch.cuda.Event(); ev.record(st); evs.append(ev)
            if len(evs) > 8:
                evs.pop(0).synchronize()
        st.synchronize()


def decode(rec, what):
    """records (n,4) uint32 -> per iteration summary"""
    out = []
    for it in np.unique(rec[:, 0]):
        r = rec[rec[:, 0] == it]
        idx = r[:, 1].astype(np.int64) * 4
        extra = ""
        if what == "heat":
            b, rem = idx // (H * W), idx % (H * W)
            yy, xx = rem // W, rem % W
            cell = b * (H // 8) * (W // 8) + (yy // 8) * (W // 8) + xx // 8
        else:
            cell = idx // 65
        cells = np.unique(cell)
        tiles = np.unique(cells // 256)
        desc = []
        for t in tiles:
            cs = cells[cells // 256 == t] % 256
            for wv in np.unique(cs // 32):
                ln = np.sort(cs[cs // 32 == wv] % 32)
                desc.append(f"tile {t} (wg {t % 256}, round {t // 256}) wave {wv} lanes {ln.min()}..{ln.max()} ({len(ln)})")
        got = r[:, 2].view(np.float32); exp = r[:, 3].view(np.float32)
        out.append(f"    iter {it}: {len(r)} float4 of {what} differ, {len(cells)} cells: " + "; ".join(desc) +
                   f" | e.g. got {got[:3]} expected {exp[:3]}" + extra)
    return out


st_a = torch.cuda.Stream()
results = []
for variant in [int(v) for v in args.variants.split(",")]:
    heat = torch.empty(B, H, W, device="cuda")
    logits = torch.empty(ncell, 65, device="cuda") if args.logits else None
    with torch.cuda.stream(st_a):
        run(variant, st_a, 1, 0, heat, None, logits, None, None, None, img=x)
        st_a.synchronize()
        heat_ref = heat.clone(); logits_ref = logits.clone() if args.logits else None
        # quiet determinism check
        rep_h = torch.zeros(4 + 4 * CAP, dtype=torch.int32, device="cuda"); rep_l = torch.zeros_like(rep_h)
        run(variant, st_a, 200, 0, heat, heat_ref, logits, logits_ref, rep_h, rep_l)
        st_a.synchronize()
    print(f"variant {variant}: quiet 200 launches: heat mismatches {int(rep_h[0])}, logits {int(rep_l[0])}; heat sum {float(heat_ref.double().sum()):.6f}", flush=True)
    for kind in args.foreign.split(","):
        rep_h.zero_(); rep_l.zero_()
        stop = threading.Event()
        th = None
        if kind != "none":
            th = threading.Thread(target=foreign_thread, args=(kind, stop), daemon=True)
            th.start()
            time.sleep(0.5)
        t0 = time.time(); done = 0; seen_h = seen_l = 0
        with torch.cuda.stream(st_a):
            while done < args.iters and time.time() - t0 < args.max_seconds:
                run(variant, st_a, args.chunk, done, heat, heat_ref, logits, logits_ref, rep_h, rep_l)
                done += args.chunk
                if (done // args.chunk) % 4 == 0:
                    st_a.synchronize()
                    nh, nl = int(rep_h[0]), int(rep_l[0])
                    if nh != seen_h or nl != seen_l:
                        seen_h, seen_l = nh, nl
            st_a.synchronize()
        dt = time.time() - t0
        stop.set()
        if th: th.join()
        nh, nl = int(rep_h[0]), int(rep_l[0])
        rh = rep_h[4:4 + 4 * min(nh, CAP)].cpu().numpy().view(np.uint32).reshape(-1, 4)
        rl = rep_l[4:4 + 4 * min(nl, CAP)].cpu().numpy().view(np.uint32).reshape(-1, 4)
        ev_h = len(np.unique(rh[:, 0])) if nh else 0
        line = f"variant {variant} foreign {kind}: {done} launches in {dt:.1f} s ({dt / done * 1e6:.0f} us each): {ev_h} launches with a wrong heat map ({nh} float4), {nl} logits float4"
        print(line, flush=True)
        results.append(line)
        for l in decode(rh, "heat")[:12]: print(l, flush=True)
        for l in decode(rl, "logits")[:12]: print(l, flush=True)
        if nl and nl <= CAP:      # which logits of a wrong cell differ (feature index histogram of the first event)
            it0 = rl[0, 0]; r0 = rl[rl[:, 0] == it0]
            feats = np.unique((r0[:, 1].astype(np.int64) * 4) % 65)
            print(f"    first event: float4 groups start at logit indices (mod 65) {feats[:40]} ... ({len(feats)} distinct)", flush=True)
print("== summary")
for l in results: print(l)
