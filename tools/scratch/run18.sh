#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for hd in f32 bx; do
(cd /tmp && XFH_HEADS=$hd timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_hd$hd" -o r02 --output-format csv -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --cpu-seconds 0 --no-side-passes > "$OLDPWD/gpurun_out/rocprof.log" 2>&1); echo "rocprof rc=$?"
done
rocm-smi --showclocks --showpower 2>/dev/null | head -20
(python bench.py --steps 300 --warmup 5 --cpu-seconds 0 --no-side-passes > /dev/null 2>&1 &) ; sleep 4; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|mclk" | head; sleep 3; rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|power" | head -4; wait
