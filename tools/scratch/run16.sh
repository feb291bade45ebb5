#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
XFH_HEADS=f32 python tools/scratch/dbg_heads.py 2>&1 | grep -v amdgpu | tail -3
XFH_HEADS=bx python tools/scratch/dbg_heads.py 2>&1 | grep -v amdgpu | tail -3
python - <<'P'
import numpy as np
a = np.load("gpurun_out/heads_f32.npz"); b = np.load("gpurun_out/heads_bx.npz")
for k in sorted(a.files):
    d = np.abs(a[k] - b[k]); i = np.unravel_index(d.argmax(), d.shape)
    print(k, a[k].shape, "max diff", d.max(), "at", i, "count > 1e-4:", int((d > 1e-4).sum()), "ref absmax", np.abs(a[k]).max())
for t in "01":
    for r in (1, 2):
        print("determinism bx", t, r, np.abs(b[f"logits{t}_0"] - b[f"logits{t}_{r}"]).max(), np.abs(b[f"rel{t}_0"] - b[f"rel{t}_{r}"]).max())
d = np.abs(a["rel1_0"] - b["rel1_0"]).reshape(-1); idx = np.nonzero(d > 1e-4)[0]; print("bad rel cells", idx[:40], len(idx))
d = np.abs(a["logits1_0"] - b["logits1_0"]); print("logits shape", d.shape)
P
