import sys, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import fixtures
from accelerated_features_amd import XFeat
xf = XFeat(weights=fixtures.synthetic_state_dict(0))
base = fixtures.texture_images(2, 2144, 3840, seed=5).cuda()
ref = xf.detectAndCompute(base, top_k=4096)
for B in (24, 128):
    x = base.repeat(B // 2, 1, 1, 1).contiguous()
    print("B", B, "image GB", x.numel() * 4 / 2**30, flush=True)
    try:
        out = xf.detectAndCompute(x, top_k=4096)
    except Exception as e:
        print("  raised:", str(e)[:200]); continue
    ok = True
    for i in (0, 1, B // 2, B - 2, B - 1):
        r = ref[i % 2]
        same = out[i]['keypoints'].shape == r['keypoints'].shape and torch.equal(out[i]['keypoints'], r['keypoints']) and torch.equal(out[i]['descriptors'], r['descriptors'])
        ok &= bool(same)
        if not same: print("  image", i, "differs", out[i]['keypoints'].shape, r['keypoints'].shape)
    print("  batch-independent:", ok, flush=True)
    del out, x
    torch.cuda.empty_cache()
