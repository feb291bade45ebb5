#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for bx in 1 3 1 3; do
XFH_BX=$bx python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-side-passes 2>&1 | grep "^{" | cut -c1-140
done
