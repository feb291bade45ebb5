"""The reference's minimal_example.py (/root/reference/minimal_example.py:1-48) against the MI355X-native drop-in:
only the import changes.  Needs an MI355X; without the trained checkpoint (weights/xfeat.pt is not redistributed
here) it falls back to the seeded synthetic weights of the test fixtures so that the flow can still be exercised.

    python examples/minimal_example.py [path/to/xfeat.pt]
"""
import os
import sys

import torch
import tqdm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from accelerated_features_amd import XFeat  # noqa: E402   (reference: from modules.xfeat import XFeat)

if len(sys.argv) > 1:
    xfeat = XFeat(weights=sys.argv[1])
else:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fixtures
    xfeat = XFeat(weights=fixtures.synthetic_state_dict(0))

# Random input
x = torch.randn(1, 3, 480, 640)

# Simple inference with batch = 1
output = xfeat.detectAndCompute(x, top_k=4096)[0]
print("----------------")
print("keypoints: ", output['keypoints'].shape)
print("descriptors: ", output['descriptors'].shape)
print("scores: ", output['scores'].shape)
print("----------------\n")

x = torch.randn(1, 3, 480, 640)
# Stress test
for i in tqdm.tqdm(range(100), desc="Stress test on VGA resolution"):
    output = xfeat.detectAndCompute(x, top_k=4096)

# Batched mode
x = torch.randn(4, 3, 480, 640)
outputs = xfeat.detectAndCompute(x, top_k=4096)
print("# detected features on each batch item:", [len(o['keypoints']) for o in outputs])

# Match two images with sparse features
x1 = torch.randn(1, 3, 480, 640)
x2 = torch.randn(1, 3, 480, 640)
mkpts_0, mkpts_1 = xfeat.match_xfeat(x1, x2)
print("match_xfeat:", mkpts_0.shape, mkpts_1.shape)

# Match two images with semi-dense approach -- batched mode with batch size 4
x1 = torch.randn(4, 3, 480, 640)
x2 = torch.randn(4, 3, 480, 640)
matches_list = xfeat.match_xfeat_star(x1, x2)
print(matches_list[0].shape)
