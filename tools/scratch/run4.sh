#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -DXFH_B1_TRACE -Wno-unused-result -Iinclude tools/bench_src/block1_bench.hip -o /tmp/block1_bench 2>/dev/null && /tmp/block1_bench 2>&1 | tee gpurun_out/block1_bench.log
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/pytest_gpu.log | grep -v amdgpu.ids | cut -c1-300
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o r02 --output-format csv -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --cpu-seconds 0 --no-side-passes > "$OLDPWD/gpurun_out/rocprof.log" 2>&1); echo "rocprof rc=$?"
for i in 1 2; do for v in 1 2; do
XFH_BLOCK1=$v python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-side-passes 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('block1 v$v fps %9.1f ms/step %.4f' % (d['value'], d['ms_per_step']))
"
done; done
