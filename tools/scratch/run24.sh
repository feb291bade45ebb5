#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu.log | grep -v amdgpu.ids | cut -c1-300
for i in 1 2 3 4; do
for bx in 9 1; do
echo -n "XFH_BX=$bx "; XFH_BX=$bx python bench.py --steps 60 --warmup 10 --cpu-seconds 0 --no-side-passes 2>&1 | grep "^{" | sed 's/.*"ms_per_step": \([0-9.]*\).*/\1/'
done
done
