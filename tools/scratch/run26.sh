#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -12 gpurun_out/pytest_gpu.log | grep -v amdgpu.ids | cut -c1-400
for i in 1 2 3; do
for m in valu c4; do
echo -n "XFH_BLOCK1=$m "; XFH_BLOCK1=$m python bench.py --steps 60 --warmup 10 --cpu-seconds 0 --no-side-passes 2>&1 | grep "^{" | sed 's/.*"ms_per_step": \([0-9.]*\).*"avg_launch_us": \([0-9.]*\).*/\1 ms  block1 \2 us/'
done
done
