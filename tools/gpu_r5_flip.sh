#!/bin/bash
# Branch r5-flip on the GPU: the defaults the round-4 probe points at (block1 = 7, fp16-pair heads: heads_f32 = 0 + fx bit 8) meet the suite, the bench and the soaks.
# Fail-fast order; every step time-boxed; logs under gpurun_out/r05_flip_*.  ~12 GPU-min without the long soaks ("nosoak"), +12 for 2 x 100 k steps.
#   before the visit (CPU):  python -m accelerated_features_amd.build && python -m accelerated_features_amd.build --scan     (the .so files travel with the snapshot)
#   gpurun --timeout 1500 -- 'bash tools/gpu_r5_flip.sh nosoak'      # first call: is the flip green and what is it worth
#   gpurun --timeout 1800 -- 'SOAK_STEPS=100000 bash tools/gpu_r5_flip.sh soakonly'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
O=gpurun_out/r05_flip; N=${SOAK_STEPS:-100000}
if [ "$1" != "soakonly" ]; then
# 1. what the flip touches, first (~1.5 min): the forms alone, the golden backbone under every option set, the range fallback (it must land on the f32 heads), the default mirrors
timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "block1_forms_alone or alternative_kernels or fp16_pair_arithmetic or conv_layers_isolated or golden or bench_shape" > ${O}_pytest_first.log 2>&1
rc=$?; echo "first tests rc=$rc"; tail -4 ${O}_pytest_first.log | grep -v amdgpu.ids
[ $rc -ne 0 ] && { grep -E "^(FAILED|E  )" ${O}_pytest_first.log | head -20; echo "STOP: the flip is not green -- go back to r5-prep's defaults (git checkout r5-prep) and read the failure against tests/emu"; exit 1; }
# 2. the whole suite (incl. the two-stream + cold-start soak over every option set, the census of the bench batch), smoke (~5 min)
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > ${O}_pytest_all.log 2>&1; echo "pytest rc=$?"; tail -4 ${O}_pytest_all.log | grep -v amdgpu.ids; grep -E "^FAILED" ${O}_pytest_all.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > ${O}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 ${O}_smoke.log
# 3. what it is worth: the default line, then round 4's mix in the same process class (same box) for the A/B (~3 min)
timeout 900 python bench.py 2>&1 | grep -v amdgpu.ids > ${O}_bench.log; tail -c 400 ${O}_bench.log
timeout 400 python tools/ab_configs.py "heads_f32=2,block1=5,fx=3" "heads_f32=2,block1=7,fx=3" "heads_f32=0,block1=5,fx=11" "heads_f32=0,block1=7,fx=11" "heads_f32=0,block1=7,fx=139" "heads_f32=0,block1=7,fx=395" "heads_f32=0,block1=7,fx=907" "heads_f32=0,block1=7,fx=1035" "heads_f32=0,block1=7,fx=1931" --spans 3,202,203 2>&1 | grep -v amdgpu.ids > ${O}_ab.log; tail -14 ${O}_ab.log
# 4. kernel stats of the flipped default (~2 min)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof2" -o it --output-format csv -- python "$OLDPWD/bench.py" --steps 10 --warmup 3 --cpu-seconds 0 --no-side-passes > "$OLDPWD/${O}_rocprof.log" 2>&1); echo "rocprof rc=$?"
find gpurun_out/prof2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} ${O}_kernel_stats.csv; rm -rf gpurun_out/prof2
fi
if [ "$1" != "nosoak" ]; then
# 5. the cold-start position scan of the new default heads with the bf16 head as this box's control (needs libxfeat_hip_scan.so) (~3 min), then the proof soaks of what would ship
if [ -f accelerated_features_amd/libxfeat_hip_scan.so ]; then
  XFH_LIB_PATH=accelerated_features_amd/libxfeat_hip_scan.so timeout 400 python tools/head_soak.py --variants $(seq -s, 1000 1015),$(seq -s, 4000 4015) --foreign none --max-seconds 5 --logits 0 2>&1 | grep -v amdgpu.ids > ${O}_head_scan.log
  echo "scan: clean positions $(grep -c ' 0 launches with a wrong' ${O}_head_scan.log) of 32"; grep "^variant" ${O}_head_scan.log | grep -v " 0 launches with a wrong" | head -20
fi
timeout 1500 python tools/final_soak.py concurrent $N 2>&1 | grep -v amdgpu.ids > ${O}_soak_concurrent.txt; echo "concurrent rc=${PIPESTATUS[0]}"; tail -2 ${O}_soak_concurrent.txt
timeout 1500 python tools/final_soak.py single $N 2>&1 | grep -v amdgpu.ids > ${O}_soak_single_stream.txt; echo "single rc=${PIPESTATUS[0]}"; tail -3 ${O}_soak_single_stream.txt
fi
