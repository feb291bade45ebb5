#!/bin/bash
# per-kernel profile of one LighterGlue pair:  gpurun -- 'bash tools/gpu_lg_prof.sh [N] [prune]'
N=${1:-4096}; P=${2:-1536}
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/lg_prof" -o lg --output-format csv -- python "$OLDPWD/tools/bench_lighterglue.py" --n $N --prune $P --iters 5 > "$OLDPWD/gpurun_out/lg_prof.log" 2>&1; echo rc=$?)
python - <<PY
import csv, glob
f = glob.glob("gpurun_out/lg_prof/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f'{r["Name"][:80]:80s} {r["Calls"]:>6s} {float(r["TotalDurationNs"])/8e6:9.3f} ms/pair  avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Percentage"]}%')
PY
