import ctypes as C, sys
import torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fixtures
from accelerated_features_amd import XFeat, _lib
xf = XFeat(weights=fixtures.synthetic_state_dict(0)); lib = _lib.load(); h = xf.net.handle()
torch.manual_seed(0)
for (hc, wc, k, B) in ((164, 164, 3276, 8), (76, 76, 819, 8), (164, 164, 3276, 1), (200, 200, 4096, 3)):
    rel = torch.rand(B, hc, wc, device="cuda")
    feats = torch.randn(B, hc, wc, 64, device="cuda")
    ref = torch.topk(rel.reshape(B, -1), k, dim=-1)[1]
    bad = 0
    for rep in range(6):
        kp = torch.empty(B, k, 2, device="cuda"); de = torch.empty(B, k, 64, device="cuda"); ci = torch.empty(B, k, dtype=torch.int32, device="cuda")
        ws, n = xf.net.workspace("dense", lib.xfh_dense_workspace_bytes(B, hc, wc, k))
        rc = lib.xfh_extract_dense(h, C.c_void_p(rel.data_ptr()), C.c_void_p(feats.data_ptr()), B, hc, wc, k, 1.0, 1.0, 1.0,
                                   C.c_void_p(kp.data_ptr()), C.c_void_p(de.data_ptr()), C.c_void_p(ci.data_ptr()), C.c_void_p(ws.data_ptr()), n, None)
        assert rc == 0
        torch.cuda.synchronize()
        eq = (ci.long() == ref).all(dim=1)
        if not bool(eq.all()):
            bad += 1
            b = int((~eq).nonzero()[0])
            nd = int((ci[b].long() != ref[b]).sum())
            print("  mismatch", (hc, wc, k, B), "rep", rep, "images", (~eq).nonzero().flatten().tolist(), "rows differing in first bad image", nd,
                  "set equal", set(ci[b].tolist()) == set(ref[b].tolist()))
    print((hc, wc, k, B), "bad reps:", bad, "/ 6")
