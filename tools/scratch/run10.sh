#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 100 python tools/bx_time.py 2>&1 | grep DBG
timeout 300 python tools/bx_check.py 2>&1 | tail -4
timeout 100 python tools/bx_time.py 2>&1 | grep DBG
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o r02 --output-format csv -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --cpu-seconds 0 --no-side-passes > "$OLDPWD/gpurun_out/rocprof.log" 2>&1); echo "rocprof rc=$?"
head -12 gpurun_out/prof/r02_kernel_stats.csv | cut -c1-150
