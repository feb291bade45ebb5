#!/usr/bin/env python3
"""The round's proof soak of what SHIPS (default options): every output of every step compared bit for bit, on the device, with the result of a quiet run.

  concurrent N   two XFeat instances on two HIP streams (FrameStream(concurrent=True)'s schedule: batches alternate, two in flight), the FULL step -- backbone, detection,
                 descriptors, fp16-filter matcher -- N steps in all; every kernel of the library runs next to every other one
  single N       ONE stream, N backbone steps (feats, heat, reliability, 1/|feats|); the second half of them with every matrix-core kernel started on an invalidated
                 instruction cache (xfh_debug_cold_start: the trigger of the round-3 / round-4 head hazard)
  python tools/final_soak.py concurrent 300000 ; python tools/final_soak.py single 300000        (logs: profiles/r04_soak_*.txt)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from accelerated_features_amd import XFeat, _lib
mode = sys.argv[1]
steps = int(sys.argv[2])
opts = list(zip(sys.argv[3::2], [int(v) for v in sys.argv[4::2]]))
lib = _lib.load()
sd = fixtures.synthetic_state_dict(0)
x = torch.cat([fixtures.texture_images(8, 480, 640, seed=77)] * 8).cuda()
ov = {}
for k in (b"fx", b"block1", b"match_exact", b"resize2"):
    import ctypes as C
    v = C.c_int(); h0 = XFeat(weights=sd, top_k=4096); lib.xfh_get_option(h0.net.handle(), k, C.byref(v)); ov[k.decode()] = v.value; del h0
print(f"final_soak {mode} {steps} steps; library defaults {ov}; overrides {opts}", flush=True)


def model():
    m = XFeat(weights=sd, top_k=4096)
    for k_, v_ in opts: m.set_option(k_, v_)
    return m


def full_step(m):
    f, _, h, r, inv = None, None, None, None, None
    kp, sc, de, nv, nc, cap, hw, d16 = m._detect_device(x, 4096, 0.05, want_f16=True)
    i0, i1, nm = m.match_pairs_device(de, nv, -1, d16)
    return kp, sc, de, i0, i1, torch.cat([nv, nc, nm])


def backbone_step(m):
    f, _, h, r, inv = m.net.backbone(x, want_logits=False, want_heat=True, want_invnorm=True)
    return f, h, r, inv


with torch.inference_mode():
    t_all = time.time()
    if mode == "concurrent":
        names = ("keypoints", "scores", "descriptors", "idx0", "idx1", "counts")
        ref = model()
        want = [t.clone() for t in full_step(ref)]
        nv = want[5][:64]; nm = want[5][128:]
        # rows beyond the counts are unspecified: compare the first `lo` rows, lo = the smallest count of the batch
        lo_k, lo_m = int(nv.min()), int(nm.min())
        torch.cuda.synchronize()
        lanes = [model(), model()]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        bad = [torch.zeros(len(names), dtype=torch.int64, device="cuda") for _ in range(2)]
        done = 0
        while done < steps:
            for _ in range(100):
                k = done & 1
                with torch.cuda.stream(streams[k]):
                    out = full_step(lanes[k])
                    for j, t in enumerate(out):
                        w = want[j]
                        if j in (0, 1, 2): d = (t[:, :lo_k] != w[:, :lo_k]).any()
                        elif j in (3, 4): d = (t[:, :lo_m] != w[:, :lo_m]).any()
                        else: d = (t != w).any()
                        bad[k][j] += d
                done += 1
            for s_ in streams: s_.synchronize()          # (bounds the queue; the two lanes stay two batches deep inside the 100)
            if done % 20000 == 0:
                tot = (bad[0] + bad[1]).tolist()
                print(f"  {done} steps, {time.time() - t_all:.0f} s: steps with a differing tensor {dict(zip(names, tot))}", flush=True)
        tot = (bad[0] + bad[1]).tolist()
        print(f"RESULT concurrent: {done} full steps on two streams in {time.time() - t_all:.0f} s ({(time.time() - t_all) / done * 1e3:.3f} ms per step): steps with a differing tensor {dict(zip(names, tot))}", flush=True)
        sys.exit(1 if any(tot) else 0)
    else:
        names = ("feats", "heat", "rel", "inv")
        m = model()
        want = [t.clone() for t in backbone_step(m)]
        torch.cuda.synchronize()
        total = {0: [0] * 4, 1: [0] * 4}
        for cold in (0, 1):
            lib.xfh_debug_cold_start(cold)
            bad = torch.zeros(4, dtype=torch.int64, device="cuda")
            n = steps // 2
            t0 = time.time(); done = 0
            while done < n:
                for _ in range(100):
                    for j, t in enumerate(backbone_step(m)):
                        bad[j] += (t != want[j]).any()
                    done += 1
                torch.cuda.synchronize()
                if done % 30000 == 0:
                    print(f"  cold_start {cold}: {done} steps, {time.time() - t0:.0f} s: {dict(zip(names, bad.tolist()))}", flush=True)
            total[cold] = bad.tolist()
            print(f"RESULT single stream, cold_start {cold}: {done} backbone steps in {time.time() - t0:.0f} s ({(time.time() - t0) / done * 1e3:.3f} ms per step): steps with a differing tensor {dict(zip(names, total[cold]))}", flush=True)
        lib.xfh_debug_cold_start(0)
        sys.exit(1 if any(total[0]) or any(total[1]) else 0)
