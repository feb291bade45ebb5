#!/bin/bash
# Bank the tree's defaults on the GPU: the suite, smoke, the default bench line, rocprofv3 kernel stats (two lanes = the default, and one lane).  ~6 GPU-min.
#   gpurun --timeout 1500 -- 'bash tools/gpu_bank.sh r05_b'      -> gpurun_out/<tag>_{pytest.log, smoke.log, bench.log, kernel_stats.csv, kernel_stats_1lane.csv}
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${1:-bank}; O=gpurun_out/$T
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > ${O}_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 ${O}_pytest.log | grep -v amdgpu.ids; grep -E "^(FAILED|ERROR)" ${O}_pytest.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > ${O}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 ${O}_smoke.log
(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temp" | head -12) > ${O}_gpu_state.txt
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids > ${O}_bench.log; tail -c 300 ${O}_bench.log
(rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temp" | head -12) >> ${O}_gpu_state.txt      # the box's state before and after the bench (bench.py records the same from sysfs: config.gpu_state)
if [ "$2" != "nostats" ]; then
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof2" -o it --output-format csv -- python "$OLDPWD/bench.py" --steps 10 --warmup 3 --cpu-seconds 0 --no-side-passes > "$OLDPWD/${O}_rocprof.log" 2>&1); echo "rocprof rc=$?"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof1" -o it --output-format csv -- python "$OLDPWD/bench.py" --steps 10 --warmup 3 --cpu-seconds 0 --no-side-passes --lanes 1 > "$OLDPWD/${O}_rocprof1.log" 2>&1); echo "rocprof 1 lane rc=$?"
find gpurun_out/prof2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} ${O}_kernel_stats.csv
find gpurun_out/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} ${O}_kernel_stats_1lane.csv
rm -rf gpurun_out/prof1 gpurun_out/prof2
fi
