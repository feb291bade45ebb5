#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_gpu.log | grep -v amdgpu.ids | cut -c1-300
for hd in f32 bx f32 bx; do
XFH_HEADS=$hd python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-side-passes 2>&1 | grep "^{" | cut -c1-140
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o r02 --output-format csv -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --cpu-seconds 0 --no-side-passes > "$OLDPWD/gpurun_out/rocprof.log" 2>&1); echo "rocprof rc=$?"
grep -i "head" gpurun_out/prof/r02_kernel_stats.csv | cut -c1-140
