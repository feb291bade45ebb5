#!/bin/bash
# per-kernel profile of a bench.py workload:  gpurun -- 'bash tools/gpu_prof_workload.sh dense [steps]'
WL=${1:-dense}; STEPS=${2:-3}
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_$WL" -o p --output-format csv -- python "$OLDPWD/bench.py" --workload $WL --steps $STEPS --warmup 1 --cpu-seconds 0 > "$OLDPWD/gpurun_out/prof_$WL.log" 2>&1; echo rc=$?)
tail -1 gpurun_out/prof_$WL.log | cut -c1-300
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_$WL/**/*kernel_stats.csv", recursive=True)[0]
n = $STEPS + 1
for r in list(csv.DictReader(open(f)))[:28]:
    print(f'{r["Name"][:95]:95s} {int(r["Calls"])/n:7.1f}/step {float(r["TotalDurationNs"])/n/1e6:8.3f} ms/step  avg {float(r["AverageNs"])/1e3:8.1f} us  {r["Percentage"]}%')
PY
