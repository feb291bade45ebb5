#!/bin/bash
# Closing visit after the fine_matcher chain: the code-position scan of the chain's kernels (control + xfh_fine_matcher + xfh_refine_matches at 16 positions),
# suite / smoke / default bench line, the dense workload's bench line + kernel stats, PMC of the chain.   gpurun --timeout 1500 -- 'bash tools/gpu_final_r5g.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 400 gpurun_probe/scan_probe accelerated_features_amd gpurun_probe/weights.bin 10000 6000 fine > gpurun_out/r05_scan_fine_chain.txt 2>&1; echo "scan rc=$?"
grep -v "^position" gpurun_out/r05_scan_fine_chain.txt | tail -8
bash tools/gpu_bank.sh r05_g nostats
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_dense" -o p --output-format csv -- python "$OLDPWD/bench.py" --workload dense --steps 3 --warmup 1 > "$OLDPWD/gpurun_out/r05_g_dense.log" 2>&1; echo "dense rc=$?")
find gpurun_out/prof_dense -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05_g_dense_kernel_stats.csv
rm -rf gpurun_out/prof_dense
grep '^{' gpurun_out/r05_g_dense.log | tail -1 | cut -c1-300
bash tools/gpu_pmc_fine.sh r05_g | tail -3
