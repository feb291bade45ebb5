"""Where the public API's step goes on the host: detectAndCompute (List[Dict]) + match_many on the bench batch, split into the device work with its one read-back and the
Python around it (views, layout checks).  gpurun: python tools/public_api_profile.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import fixtures  # noqa: E402
from accelerated_features_amd import XFeat  # noqa: E402

B, H, W, TOP_K = 64, 480, 640, 4096
xf = XFeat(weights=fixtures.synthetic_state_dict(0), top_k=TOP_K, detection_threshold=0.05)
x = torch.cat([fixtures.texture_images(8, H, W, seed=100 + i) for i in range(8)]).cuda()


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, r


ms_dc, res = t(lambda: xf.detectAndCompute(x, top_k=TOP_K))
ms_pad, _ = t(lambda: xf.detectAndComputePadded(x, top_k=TOP_K)["n_valid"].cpu())
f1, f2 = [r["descriptors"] for r in res[0::2]], [r["descriptors"] for r in res[1::2]]
ms_mm, _ = t(lambda: xf.match_many(f1, f2, min_cossim=-1))
ms_lists, _ = t(lambda: ([r["descriptors"] for r in res[0::2]], [r["descriptors"] for r in res[1::2]]))
ms_lay, _ = t(lambda: xf._strided_layout(f1, f2))


def both():
    r = xf.detectAndCompute(x, top_k=TOP_K)
    return xf.match_many([q["descriptors"] for q in r[0::2]], [q["descriptors"] for q in r[1::2]], min_cossim=-1)


ms_both, _ = t(both)
print(f"detectAndCompute {ms_dc:.3f} ms  (padded form + its read-back {ms_pad:.3f} ms: the List[Dict] costs {ms_dc - ms_pad:.3f} ms)")
print(f"match_many {ms_mm:.3f} ms  (list comprehension {ms_lists:.3f}, _strided_layout {ms_lay:.3f})")
print(f"detectAndCompute + match_many {ms_both:.3f} ms = {B / ms_both:.1f} k frames/s;  counts: {sorted(set(int(r['keypoints'].shape[0]) for r in res))[:5]}")
