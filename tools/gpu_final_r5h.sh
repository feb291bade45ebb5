#!/bin/bash
# The round's last visit (after the dense path's staged resize): suite / smoke / default bench line, the dense workload's bench line + kernel stats.   gpurun --timeout 900 -- 'bash tools/gpu_final_r5h.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/gpu_bank.sh r05_h nostats
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_dense" -o p --output-format csv -- python "$OLDPWD/bench.py" --workload dense --steps 3 --warmup 1 --cpu-seconds 0 > "$OLDPWD/gpurun_out/r05_h_dense.log" 2>&1; echo "dense rc=$?")
find gpurun_out/prof_dense -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05_h_dense_kernel_stats.csv
rm -rf gpurun_out/prof_dense
grep '^{' gpurun_out/r05_h_dense.log | tail -1 | cut -c1-200
