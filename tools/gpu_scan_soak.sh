#!/bin/bash
# Round 5's torture evidence in one visit: the code-position scan (tools/bench_src/scan_probe.cpp; no Python), then the proof soaks of the
# shipped defaults (tools/final_soak.py).  Before the visit (CPU): python -m accelerated_features_amd.build [--shift 1..15], tools/dump_weights.py, hipcc scan_probe.
#   gpurun --timeout 1700 -- 'N=50000 SOAK=150000 bash tools/gpu_scan_soak.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
N=${N:-50000}; SOAK=${SOAK:-150000}; TAG=${TAG:-r05}
timeout 900 gpurun_probe/scan_probe accelerated_features_amd gpurun_probe/weights.bin $N > gpurun_out/${TAG}_scan_all_kernels.txt 2>&1; echo "scan rc=$?"
grep -v "^position" gpurun_out/${TAG}_scan_all_kernels.txt | tail -20
if [ "$SOAK" != "0" ]; then
timeout 700 python tools/final_soak.py concurrent $SOAK 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_soak_concurrent.txt; echo "concurrent rc=${PIPESTATUS[0]}"; tail -2 gpurun_out/${TAG}_soak_concurrent.txt
timeout 700 python tools/final_soak.py single $SOAK 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_soak_single_stream.txt; echo "single rc=${PIPESTATUS[0]}"; tail -3 gpurun_out/${TAG}_soak_single_stream.txt
fi
