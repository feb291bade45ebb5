#!/bin/bash
# One GPU visit: parity suite, bench, rocprof kernel stats.   tools/gpu_visit.sh <tag> [pytest-args...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-visit}; shift
mkdir -p gpurun_out/$TAG
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider "$@" > gpurun_out/$TAG/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/$TAG/pytest_gpu.log | grep -v amdgpu.ids
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$TAG/bench.log | tail -3
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/$TAG/prof" -o it -- python "$OLDPWD/bench.py" --steps 10 --warmup 3 --cpu-seconds 0 > "$OLDPWD/gpurun_out/$TAG/rocprof.log" 2>&1); echo "rocprof rc=$?"
find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/$TAG/kernel_stats.csv
head -32 gpurun_out/$TAG/kernel_stats.csv | cut -c1-200
find gpurun_out/$TAG/prof -type f ! -name "*stats*" -delete
