"""A/B of option resize2 (the fused two-stage resize of the dual-scale dense path: 1 = input region staged in LDS, 0 = four-byte gathers): bit-identical outputs of
extract_dualscale on 1024^2 images, HIP-event time of xfh_backbone_resized's first span per scale, and of the whole match_xfeat_star step."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fixtures
from accelerated_features_amd import XFeat
xf = XFeat(weights=fixtures.synthetic_state_dict(0), top_k=4096)
x = fixtures.texture_images(16, 1024, 1024, seed=5).cuda()
outs = {}
for v in (0, 1, 0, 1):
    xf.net.set_option('resize2', v)
    for _ in range(2): r = xf.extract_dualscale(x, 10000)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): r = xf.extract_dualscale(x, 10000)
    e1.record(); torch.cuda.synchronize()
    print(f"resize2 = {v}: extract_dualscale(16 x 1024^2) {e0.elapsed_time(e1) / 5:.3f} ms")
    outs.setdefault(v, r)
same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
print('outputs bit-identical:', same)
assert same
