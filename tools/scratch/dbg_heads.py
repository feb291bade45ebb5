import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
from accelerated_features_amd import XFeat
g = np.load(os.path.join(ROOT, "tests/golden/g5_assets.npz"))
xf = XFeat(weights=fixtures.synthetic_state_dict(0), top_k=4096, detection_threshold=0.05)
out = {}
for t in ("0", "1"):
    x, rh, rw = xf.preprocess_tensor(xf.parse_input(g["img" + t]))
    for rep in range(3):
        feats, logits, rel = xf.net(x.cuda())
        out[f"logits{t}_{rep}"] = logits.cpu().numpy(); out[f"rel{t}_{rep}"] = rel.cpu().numpy(); out[f"feats{t}"] = feats.cpu().numpy()
np.savez(os.path.join(ROOT, "gpurun_out", f"heads_{os.environ.get('XFH_HEADS', 'bx')}.npz"), **out)
