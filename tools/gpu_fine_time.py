"""Time the fine_matcher MLP (modules/model.py:97-111) alone: xfh_fine_matcher on n rows, HIP events.  XFH_LIB_PATH selects an A/B build.
usage: python tools/gpu_fine_time.py [n_rows]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import fixtures
from accelerated_features_amd import XFeat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
xf = XFeat(weights=fixtures.synthetic_state_dict(0))
x = torch.randn(n, 128, device='cuda')
for _ in range(3): xf.net._fine_matcher(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): xf.net._fine_matcher(x)
e1.record(); torch.cuda.synchronize()
print(f"fine_matcher n={n}: {e0.elapsed_time(e1) / 20 * 1000:.1f} us")
