#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -DXFH_B1_TRACE -Wno-unused-result -Iinclude tools/bench_src/block1_bench.hip -o /tmp/block1_bench 2>/dev/null && /tmp/block1_bench 2>&1 | tee gpurun_out/block1_bench.log
for dbg in 0 1 2 3 4 8 12; do
(cd /tmp && XFH_TOPK_DBG=$dbg timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_dbg$dbg" -o t --output-format csv -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --cpu-seconds 0 --no-side-passes > /dev/null 2>&1)
echo "dbg=$dbg: $(grep -h "topk_" gpurun_out/prof_dbg$dbg/*kernel_stats.csv | awk -F, '{printf "%s %s us | ", substr($1,1,40), $4/1000}')"
done
