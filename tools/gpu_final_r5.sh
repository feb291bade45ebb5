#!/bin/bash
# Round 5's closing evidence: suite, smoke, default bench line, rocprofv3 kernel stats (two lanes, one lane), PMC traffic, and the other BASELINE workloads
# (dense = configs[2], lighterglue = configs[4], megadepth = configs[3] on one GPU): bench line + kernel stats each.   gpurun --timeout 2400 -- 'bash tools/gpu_final_r5.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
bash tools/gpu_bank.sh r05_final
bash tools/gpu_traffic.sh > gpurun_out/r05_final_traffic.log 2>&1; tail -34 gpurun_out/r05_final_traffic.log
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
for WL in dense lighterglue megadepth; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_$WL" -o p --output-format csv -- python "$OLDPWD/bench.py" --workload $WL --steps 3 --warmup 1 > "$OLDPWD/gpurun_out/r05_final_${WL}.log" 2>&1; echo "$WL rc=$?")
  find gpurun_out/prof_$WL -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05_final_${WL}_kernel_stats.csv
  rm -rf gpurun_out/prof_$WL
  grep '^{' gpurun_out/r05_final_${WL}.log | tail -1 | cut -c1-400
done
