#!/bin/bash
# One GPU visit: parity suite, bench, rocprof kernel stats.   tools/gpu_visit.sh <tag> [pytest-args...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-visit}; shift
mkdir -p gpurun_out/$TAG
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider "$@" > gpurun_out/$TAG/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/$TAG/pytest_gpu.log | grep -v amdgpu.ids
timeout 600 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-side-passes 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$TAG/bench.log | tail -3
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/$TAG/prof" -o it --output-format csv -- python "$OLDPWD/bench.py" --steps 10 --warmup 3 --cpu-seconds 0 --no-side-passes > "$OLDPWD/gpurun_out/$TAG/rocprof.log" 2>&1); echo "rocprof rc=$?"
find gpurun_out/$TAG/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/$TAG/kernel_stats.csv
# the same command with one lane: per-kernel durations without another batch's kernels on the chip (the table printed below)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/$TAG/prof1" -o it --output-format csv -- python "$OLDPWD/bench.py" --steps 10 --warmup 3 --cpu-seconds 0 --no-side-passes --lanes 1 > "$OLDPWD/gpurun_out/$TAG/rocprof_1lane.log" 2>&1); echo "rocprof (1 lane) rc=$?"
cp gpurun_out/$TAG/kernel_stats.csv gpurun_out/$TAG/kernel_stats_2lanes.csv
find gpurun_out/$TAG/prof1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/$TAG/kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/$TAG/kernel_stats.csv")))
n = 14.0       # steps + warmup + side passes of the profiled command: 10 + 3 + 1 (arm)...; per-step figures use Calls of block1
calls = next((int(r["Calls"]) for r in rows if "block1" in r["Name"]), 1)
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step (block1 launches = {calls}): {tot / calls / 1e3:.1f} us")
for r in rows[:34]:
    print(f'{r["Name"].replace("xfh::", "").replace("void ", "")[:70]:70s} {int(r["Calls"]) / calls:5.1f}/step {float(r["TotalDurationNs"]) / calls / 1e3:8.1f} us/step  avg {float(r["AverageNs"]) / 1e3:8.1f} us  {r["Percentage"]}%')
PY
rm -rf gpurun_out/$TAG/prof gpurun_out/$TAG/prof1
