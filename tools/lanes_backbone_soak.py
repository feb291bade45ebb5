#!/usr/bin/env python3
"""Two lanes running ONLY the backbone (feats, heat map, reliability, 1/|feats|) of the bench batch concurrently, every output of every step compared bit for bit with
a reference: which tensor moves first when a two-lane step goes wrong, and where.   python tools/lanes_backbone_soak.py [steps] [recreate-every]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from accelerated_features_amd import XFeat
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
every = int(sys.argv[2]) if len(sys.argv) > 2 else 40
NL = int(os.environ.get('SOAK_LANES', '2'))          # 1: the same loop on ONE stream (is a rare difference a matter of concurrency at all?)
opts = list(zip(sys.argv[3::2], [int(v) for v in sys.argv[4::2]]))      # per-handle options for the reference and the lanes, e.g. heads_f32 1
sd = fixtures.synthetic_state_dict(0)
x = torch.cat([fixtures.texture_images(8, 480, 640, seed=77)] * 8).cuda()
ref = XFeat(weights=sd, top_k=4096)
for k_, v_ in opts: ref.set_option(k_, v_)
names = ("feats", "heat", "rel", "inv")
with torch.inference_mode():
    f0, _, h0, r0, v0 = ref.net.backbone(x, want_logits=False, want_heat=True, want_invnorm=True)
want = dict(zip(names, (f0, h0, r0, v0)))
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(2)]
lanes = None
pending = [None, None]
nbad = 0


def check(k, step):
    global nbad
    out, ev = pending[k]
    ev.synchronize()
    for n, t in zip(names, out):
        if not torch.equal(t, want[n]):
            d = (t != want[n])
            imgs = sorted(set(d.flatten(1).any(1).nonzero().flatten().tolist()))
            md = float((t - want[n]).abs().max())
            print(f"step {step} lane {k}: {n} differs: {int(d.sum())} values, max |diff| {md:.3g}, images {imgs}", flush=True)
            nbad += 1


with torch.inference_mode():
    for step in range(steps):
        if step % every == 0:
            for k in range(2):
                if pending[k] is not None: check(k, step); pending[k] = None
            lanes = [XFeat(weights=sd, top_k=4096) for _ in range(2)]
            for ln in lanes:
                for k_, v_ in opts: ln.set_option(k_, v_)
        k = step % NL
        if pending[k] is not None: check(k, step)
        with torch.cuda.stream(streams[k]):
            f, _, h, r, v = lanes[k].net.backbone(x, want_logits=False, want_heat=True, want_invnorm=True)
            ev = torch.cuda.Event(); ev.record(streams[k])
        pending[k] = ((f, h, r, v), ev)
    for k in range(2):
        if pending[k] is not None: check(k, steps)
print(f"{steps} concurrent backbone steps (fresh handles every {every}, options {opts}): {nbad} differing tensors")
