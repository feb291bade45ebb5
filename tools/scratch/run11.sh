#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for bx in 0 1 0 1; do
XFH_BX=$bx python bench.py --steps 30 --warmup 5 --cpu-seconds 0 --no-side-passes 2>&1 | grep "^{" | cut -c1-160
done
for bx in 0 1; do
(cd /tmp && XFH_BX=$bx timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_bx$bx" -o r02 --output-format csv -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --cpu-seconds 0 --no-side-passes > "$OLDPWD/gpurun_out/rocprof.log" 2>&1); echo "rocprof rc=$?"
done
