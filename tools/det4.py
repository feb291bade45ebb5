import ctypes as C, sys
import torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fixtures
from accelerated_features_amd import XFeat, _lib
xf = XFeat(weights=fixtures.synthetic_state_dict(0)); lib = _lib.load(); h = xf.net.handle()
base = fixtures.texture_images(2, 1024, 1024, seed=55)
a = torch.cat([base, base.flip(3), base.flip(2), base.flip(2).flip(3)]).cuda()
x = torch.nn.functional.interpolate(a, size=(1312, 1312), mode="bilinear").contiguous()
feats, logits, heat, rel = xf.net.backbone(x, True, False)
torch.cuda.synchronize()
B, hc, wc = rel.shape; k = 3276
r = rel.reshape(B, -1)
print("rel stats: max", float(r.max()), "count ==max per image", [(int((r[b] == r[b].max()).sum())) for b in range(B)], "unique vals img0", int(torch.unique(r[0]).numel()))
ref_sorted = torch.sort(r, dim=1, descending=True, stable=True)[1][:, :k]
outs = []
for rep in range(4):
    kp = torch.empty(B, k, 2, device="cuda"); de = torch.empty(B, k, 64, device="cuda"); ci = torch.full((B, k), -1, dtype=torch.int32, device="cuda")
    ws, n = xf.net.workspace("dense", lib.xfh_dense_workspace_bytes(B, hc, wc, k))
    rc = lib.xfh_extract_dense(h, C.c_void_p(rel.data_ptr()), C.c_void_p(feats.data_ptr()), B, hc, wc, k, 1.0, 1.0, 1.0,
                               C.c_void_p(kp.data_ptr()), C.c_void_p(de.data_ptr()), C.c_void_p(ci.data_ptr()), C.c_void_p(ws.data_ptr()), n, None)
    torch.cuda.synchronize()
    outs.append(ci.clone())
    eq = (ci.long() == ref_sorted).all(1)
    print("rep", rep, "equal to stable sort per image", eq.tolist(), "vs rep0", (ci == outs[0]).all(1).tolist())
    if not bool(eq.all()):
        b = int((~eq).nonzero()[0]); d = (ci[b].long() != ref_sorted[b]).nonzero().flatten()
        print("   first diffs at ranks", d[:8].tolist(), "got", ci[b][d[:8]].tolist(), "want", ref_sorted[b][d[:8]].tolist(),
              "vals got", r[b][ci[b][d[:8]].long()].tolist(), "vals want", r[b][ref_sorted[b][d[:8]]].tolist())
