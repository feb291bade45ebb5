#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/pytest_gpu.log | grep -v amdgpu.ids | cut -c1-300
python - <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import fixtures
from accelerated_features_amd import XFeat
xf = XFeat(weights=fixtures.synthetic_state_dict(0))
for shape in ((3, 96, 160), (2, 480, 640), (1, 224, 352)):
    x = fixtures.texture_images(*shape, seed=3).cuda()
    os.environ.pop("XFH_BLOCK1", None)
    a = xf.net(x)
    os.environ["XFH_BLOCK1"] = "valu"
    b = xf.net(x)
    os.environ.pop("XFH_BLOCK1", None)
    print(shape, "mfma vs valu block1: feats equal", torch.equal(a[0], b[0]), "logits", torch.equal(a[1], b[1]), "rel", torch.equal(a[2], b[2]))
PY
for i in 1 2; do for v in valu mfma; do
XFH_BLOCK1=$v python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-side-passes 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('block1 $v fps %9.1f ms/step %.4f block1 %.1f us' % (d['value'], d['ms_per_step'], d['roofline_block1']['us_per_step']))
"
done; done
