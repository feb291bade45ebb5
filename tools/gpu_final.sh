#!/bin/bash
# The round's final GPU sequence: proof soaks of the shipped defaults, the GPU test-suite, the default bench line, rocprofv3 kernel stats (default + one lane), PMC traffic.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
N=${SOAK_STEPS:-300000}; TAG=${TAG:-r05}      # (round 4's logs: profiles/r04_soak_*.txt)
timeout 120 python tools/final_soak.py concurrent 600 > gpurun_out/soak_sanity.log 2>&1 && timeout 120 python tools/final_soak.py single 600 >> gpurun_out/soak_sanity.log 2>&1; echo "sanity rc=$?"; grep RESULT gpurun_out/soak_sanity.log
if [ "$1" != "nosoak" ]; then
timeout 1500 python tools/final_soak.py concurrent $N 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_soak_concurrent.txt; echo "concurrent rc=${PIPESTATUS[0]}"; tail -2 gpurun_out/${TAG}_soak_concurrent.txt
timeout 1500 python tools/final_soak.py single $N 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_soak_single_stream.txt; echo "single rc=${PIPESTATUS[0]}"; tail -3 gpurun_out/${TAG}_soak_single_stream.txt
fi
if [ "$1" != "soakonly" ]; then
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log | grep -v amdgpu.ids
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/bench_final.log; tail -c 400 gpurun_out/bench_final.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof2" -o it --output-format csv -- python "$OLDPWD/bench.py" --steps 10 --warmup 3 --cpu-seconds 0 --no-side-passes > "$OLDPWD/gpurun_out/rocprof2.log" 2>&1); echo "rocprof rc=$?"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof1" -o it --output-format csv -- python "$OLDPWD/bench.py" --steps 10 --warmup 3 --cpu-seconds 0 --no-side-passes --lanes 1 > "$OLDPWD/gpurun_out/rocprof1.log" 2>&1); echo "rocprof 1 lane rc=$?"
find gpurun_out/prof2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_default.csv
find gpurun_out/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kernel_stats_1lane.csv
rm -rf gpurun_out/prof1 gpurun_out/prof2
bash tools/gpu_traffic.sh > gpurun_out/traffic.log 2>&1; tail -30 gpurun_out/traffic.log
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
fi
