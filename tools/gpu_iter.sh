#!/bin/bash
# Quick iteration visit: GPU parity tests + per-kernel microbench + short bench (no CPU baseline).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/pytest_gpu.log | grep -v amdgpu.ids
timeout 300 python tools/microbench.py all 2>&1 | grep -v amdgpu.ids | tee gpurun_out/microbench.log
timeout 120 python tools/trace_conv.py block3.1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/trace.log
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-seconds 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/bench.log
if [ "$1" == "prof" ]; then
(cd /tmp && timeout 900 rocprofv3 --kernel-trace -d "$OLDPWD/gpurun_out/prof" -o it -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --cpu-seconds 0 > "$OLDPWD/gpurun_out/rocprof.log" 2>&1); echo "rocprof rc=$?"
python tools/prof_summary.py gpurun_out/prof 7 | head -45
fi
