#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3 4 5; do
for hd in f32 bx; do
echo -n "$hd "; XFH_HEADS=$hd python bench.py --steps 60 --warmup 10 --cpu-seconds 0 --no-side-passes 2>&1 | grep "^{" | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1/'
done
done
