#!/bin/bash
# One GPU-box visit: build check, GPU parity tests, smoke, short bench, kernel-trace profile.
# Everything is logged under gpurun_out/ (merged back by gpurun).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
nproc >> gpurun_out/gpu.txt
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -40 gpurun_out/pytest_gpu.log
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.log
if [ "$1" != "nobench" ]; then
echo "== bench"; timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench rc=$?"; tail -3 gpurun_out/bench.log
echo "== rocprof"; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o r02 --output-format csv -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --cpu-seconds 0 > "$OLDPWD/gpurun_out/rocprof.log" 2>&1); echo "rocprof rc=$?"
find gpurun_out/prof -name "*stats*" | head
fi
