#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/pytest_gpu.log | grep -v amdgpu.ids | cut -c1-400
for m in bf16 f32 bf16 f32; do
XFH_MATCH=$m python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-side-passes 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('match $m fps %9.1f ms/step %.4f matches %.1f' % (d['value'], d['ms_per_step'], d['config']['mean_matches']))
"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o r02 --output-format csv -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --cpu-seconds 0 --no-side-passes > "$OLDPWD/gpurun_out/rocprof.log" 2>&1); echo "rocprof rc=$?"
