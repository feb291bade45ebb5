"""The XFeat branch of the reference's realtime_demo.py (/root/reference/realtime_demo.py:136-142, 204-231) without camera or window:
a reference frame is set once, every following frame is matched against its cached features and a homography is fitted -- all on the
MI355X (detectAndCompute, mutual-NN match, MAGSAC++), one read-back per frame for the print-out.

    python examples/demo_headless.py [path/to/xfeat.pt]

Without the trained checkpoint the seeded synthetic weights of the test fixtures are used; their descriptors only survive image shifts by
multiples of the backbone's stride (32 px) and do not reach the demo's min_cossim of 0.82, so the synthetic run matches with min_cossim = -1.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from accelerated_features_amd import XFeat  # noqa: E402
from accelerated_features_amd.homography import ReferenceTracker  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures  # noqa: E402

trained = len(sys.argv) > 1
xfeat = XFeat(weights=sys.argv[1]) if trained else XFeat(weights=fixtures.synthetic_state_dict(0))
tracker = ReferenceTracker(xfeat, top_k=4096, min_cossim=0.82 if trained else -1, ransac_thr=4.0, min_inliers=50)

ref = fixtures.texture_images(1, 480, 640, seed=5)                      # the demo's frame grabber, replaced by a moving synthetic texture
tracker.set_reference(ref.cuda())
rs = np.random.RandomState(0)
for i in range(1, 9):
    dy, dx = 32 * (i % 3), 32 * i
    frame = (torch.roll(ref, shifts=(dy, dx), dims=(2, 3)) + torch.from_numpy((0.01 * rs.randn(*ref.shape)).astype(np.float32))).cuda()
    t0 = time.perf_counter()
    r = tracker.track(frame)
    H, valid, info = r["H"][0].cpu().numpy(), bool(r["valid"][0]), r["info"][0].cpu().tolist()
    ms = 1e3 * (time.perf_counter() - t0)
    print(f"frame {i}: true shift ({dx:3d},{dy:3d})  matches {int(r['n_matches'][0]):4d}  inliers {info[3]:4d}  "
          f"H translation ({H[0, 2]:7.2f},{H[1, 2]:6.2f})  {'ok ' if valid else 'H = None'}  {ms:6.2f} ms")
