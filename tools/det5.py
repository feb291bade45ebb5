"""Read-before-write detector: poison every workspace with NaNs / different garbage before a call; outputs must not change."""
import sys
import torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fixtures
from accelerated_features_amd import XFeat
xf = XFeat(weights=fixtures.synthetic_state_dict(0))

def poison(val):
    for name, t in xf.net._ws.items():
        t.view(torch.uint8).fill_(val)

def run_sparse(x):
    kp, sc, de, nv, nc, cap, hw = xf._detect_device(x, 4096, 0.05)
    i0, i1, nm = xf.match_pairs_device(de, nv, -1)
    n = nm.cpu().tolist()
    return [kp, sc, de, nv, nc, nm] + [i0[p, :n[p]] for p in range(len(n))] + [i1[p, :n[p]] for p in range(len(n))]

def run_dense(a):
    d = xf.detectAndComputeDense(a, top_k=4096)
    return [d["keypoints"], d["descriptors"], d["scales"]]

def run_backbone(x):
    return [t for t in xf.net.backbone(x, True, True)]

cases = {
    "sparse VGA B=8": (run_sparse, fixtures.texture_images(8, 480, 640, seed=5).cuda()),
    "backbone 1312 B=2": (run_backbone, torch.nn.functional.interpolate(fixtures.texture_images(2, 1024, 1024, seed=55), size=(1312, 1312), mode="bilinear").cuda().contiguous()),
    "backbone 608 B=2": (run_backbone, torch.nn.functional.interpolate(fixtures.texture_images(2, 1024, 1024, seed=55), size=(608, 608), mode="bilinear").cuda().contiguous()),
    "dense 1024 B=2": (run_dense, fixtures.texture_images(2, 1024, 1024, seed=55).cuda()),
}
for name, (fn, x) in cases.items():
    fn(x); torch.cuda.synchronize()          # allocate workspaces
    res = []
    for val in (0x00, 0xFF, 0x7F, 0x00):
        poison(val); torch.cuda.synchronize()
        res.append([t.clone() for t in fn(x)]); torch.cuda.synchronize()
    ok = [[bool(torch.equal(a, b)) or (a.dtype.is_floating_point and bool(torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)))) for a, b in zip(res[0], r)] for r in res[1:]]
    print(name, "outputs equal to zero-poisoned run:", ok)
