import os, sys
import torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fixtures
from accelerated_features_amd import XFeat
xf = XFeat(weights=fixtures.synthetic_state_dict(0))
base = fixtures.texture_images(2, 1024, 1024, seed=55)
a = torch.cat([base, base.flip(3), base.flip(2), base.flip(2).flip(3)]).cuda()
outs = [xf.detectAndComputeDense(a, top_k=4096) for _ in range(4)]
torch.cuda.synchronize()
for i in range(1, 4):
    dk = (outs[i]["keypoints"] != outs[0]["keypoints"]).any(-1)
    dd = (outs[i]["descriptors"] != outs[0]["descriptors"]).any(-1)
    print(i, "kp rows differing per image", dk.sum(1).tolist(), "desc rows", dd.sum(1).tolist(),
          "first-scale part", int(dk[:, :819].sum()), "second-scale part", int(dk[:, 819:].sum()),
          "max desc diff", float((outs[i]["descriptors"] - outs[0]["descriptors"]).abs().max()))
# isolate: backbone at both scales twice
for s in (608, 1312):
    x = torch.nn.functional.interpolate(a, size=(s, s), mode="bilinear")
    f = [xf.net.backbone(x.contiguous(), True, False) for _ in range(3)]
    torch.cuda.synchronize()
    print(s, "backbone equal", [bool(torch.equal(f[0][j], f[k][j])) for k in (1, 2) for j in (0, 1, 3)])
    ed = [xf.extractDense(x.contiguous(), 3276) for _ in range(3)]
    print(s, "extractDense equal", [bool(torch.equal(ed[0][j], ed[k][j])) for k in (1, 2) for j in (0, 1)])
