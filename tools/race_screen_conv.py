import ctypes as C, sys
import torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fixtures
from accelerated_features_amd import XFeat, _lib
from accelerated_features_amd.spec import CONVS, CONV_INDEX
xf = XFeat(weights=fixtures.synthetic_state_dict(0)); lib = _lib.load(); h = xf.net.handle()
def rep_test(name, B, hin, win, reps=60):
    c = next(c for c in CONVS if c.name == name)
    hout, wout = (hin - 1) // c.stride + 1, (win - 1) // c.stride + 1
    xin = torch.randn(B, c.cin, hin, win, device="cuda")
    gen = torch.empty(B, c.cout, hout, wout, device="cuda")
    assert lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(xin.data_ptr()), B, hin, win, C.c_void_p(gen.data_ptr()), 1, None) == 0   # generic reference
    nbad = 0
    for rep in range(reps):
        y = torch.full((B, c.cout, hout, wout), float("nan"), device="cuda")
        assert lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(xin.data_ptr()), B, hin, win, C.c_void_p(y.data_ptr()), 0, None) == 0
        err = (y - gen).abs()
        wrong = ~(err <= 1e-3)
        if wrong.any():
            nbad += 1
            idx = wrong.nonzero()
            if nbad <= 3:
                print("   ", name, "rep", rep, "wrong elems", int(wrong.sum()), "b", idx[:, 0].unique().tolist()[:8], "co", idx[:, 1].unique().tolist()[:40],
                      "y range", int(idx[:, 2].min()), int(idx[:, 2].max()), "x range", int(idx[:, 3].min()), int(idx[:, 3].max()), "max err", float(err[wrong].nan_to_num(1e9).max()))
    print(name, (B, hin, win), "wrong reps:", nbad, "/", reps)
rep_test("block2.1", 8, 328, 328)
rep_test("block2.0", 8, 328, 328)
rep_test("block2.1", 64, 120, 160)
rep_test("block3.1", 8, 164, 164)
rep_test("block_fusion.0", 64, 60, 80)
rep_test("block4.1", 64, 30, 40)
rep_test("block3.0", 8, 328, 328)
