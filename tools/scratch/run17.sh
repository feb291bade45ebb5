#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/scratch/run16.sh 2>&1 | grep -v "^rel\|feats" | tail -12
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/pytest_gpu.log | grep -v amdgpu.ids | cut -c1-300
for hd in f32 bx f32 bx; do
XFH_HEADS=$hd python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-side-passes 2>&1 | grep "^{" | cut -c1-200 | sed 's/.*"value"/value/'
done
