import sys
import torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fixtures
from accelerated_features_amd import XFeat
xf = XFeat(weights=fixtures.synthetic_state_dict(0))
base = fixtures.texture_images(2, 1024, 1024, seed=55)
a = torch.cat([base, base.flip(3), base.flip(2), base.flip(2).flip(3)]).cuda()
x = torch.nn.functional.interpolate(a, size=(1312, 1312), mode="bilinear").contiguous()
def eqs(ed):
    return [[bool(torch.equal(ed[i][j], ed[k][j])) for j in (0, 1)] for i, k in ((0, 1), (0, 2), (1, 2), (2, 3))]
for mode in ("sync", "nosync", "nosync", "sync"):
    ed = []
    for _ in range(4):
        ed.append(xf.extractDense(x, 3276))
        if mode == "sync": torch.cuda.synchronize()
    torch.cuda.synchronize()
    print(mode, "pairs (0,1),(0,2),(1,2),(2,3):", eqs(ed))
    kp = [e[0] for e in ed]
    for i in range(1, 4):
        d = (kp[i] != kp[0]).any(-1)
        if d.any(): print("   call", i, "differs from call 0 in images", d.any(1).nonzero().flatten().tolist(), "rows", d.sum(1).tolist())
