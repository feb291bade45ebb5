#!/bin/bash
# First GPU visit of round 5: the kernels prepared on CPU in round 4 (block1 on the fp16 matrix cores: option block1 = 6 / 7; the fp16-pair heads: heads_f32 = 0 + fx
# bit 8; range tracking on the high parts) meet the hardware.  Order: cheapest decisive checks first; every step time-boxed; logs under gpurun_out/r05_first_*.
#   before the visit (CPU):  python -m accelerated_features_amd.build && python -m accelerated_features_amd.build --scan
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
O=gpurun_out/r05_first
# 0. two hardware assumptions of the CPU-prepared kernels (partial-exec LDS-DMA, the 16x16x32 operand layout): milliseconds
/opt/rocm/bin/hipcc -O2 -w --offload-arch=gfx950 tools/bench_src/hw_semantics.hip -o /tmp/hw_semantics 2>/dev/null && timeout 60 /tmp/hw_semantics | tee ${O}_hw_semantics.log
# 1. parity of the new forms alone and in the backbone (~1 min)
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "block1_forms_alone or alternative_kernels or fp16_pair_arithmetic or conv_layers_isolated" > ${O}_pytest_new.log 2>&1
echo "new-kernel tests rc=$?"; tail -5 ${O}_pytest_new.log | grep -v amdgpu.ids
# 2. what they are worth (in-run A/B, one process): block1 forms; heads; ~2 min
timeout 300 python tools/ab_configs.py "block1=0" "block1=6" "block1=7" --spans 3 2>&1 | grep -v amdgpu.ids > ${O}_ab_block1.log; tail -12 ${O}_ab_block1.log
timeout 400 python tools/ab_configs.py "heads_f32=3" "heads_f32=2" "heads_f32=0" "heads_f32=0,fx=11" "heads_f32=0,fx=27" "heads_f32=0,fx=43" --spans 202,203 2>&1 | grep -v amdgpu.ids > ${O}_ab_heads.log; tail -12 ${O}_ab_heads.log
timeout 300 python tools/ab_configs.py "fx=3" "fx=7" "fx=67" --spans 108,111,112,117,118 2>&1 | grep -v amdgpu.ids > ${O}_ab_bx64.log; tail -8 ${O}_ab_bx64.log      # (spans: 100 + index in spec.CONVS: block3.1 (+3.2 fused), block4.1, block4.2, block_fusion.0, block_fusion.1 (+.2 fused))
# 3. the fp16-pair head under the cold-start torture: 16 code positions next to the bf16 head as the box's control (needs libxfeat_hip_scan.so); ~4 min
if [ -f accelerated_features_amd/libxfeat_hip_scan.so ]; then
  XFH_LIB_PATH=accelerated_features_amd/libxfeat_hip_scan.so timeout 600 python tools/head_soak.py --variants $(seq -s, 1000 1015),$(seq -s, 4000 4015),$(seq -s, 5000 5015) --foreign none --max-seconds 6 --logits 0 2>&1 | grep -v amdgpu.ids > ${O}_head_scan.log
  grep -c "0 launches with a wrong" ${O}_head_scan.log; grep "^variant" ${O}_head_scan.log | grep -v " 0 launches with a wrong" | head -20
fi
# 4. two streams + cold start with the new forms on (the suite's soak, all parameters) ~2 min
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "two_streams_and_cold" -s 2>&1 | grep -v amdgpu.ids > ${O}_soak.log; grep -E "options|passed|failed" ${O}_soak.log | cut -c1-300
# 5. the whole suite + bench with the library defaults (~4 min)
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > ${O}_pytest_all.log 2>&1; echo "pytest rc=$?"; tail -3 ${O}_pytest_all.log | grep -v amdgpu.ids
timeout 600 python bench.py 2>&1 | grep -v amdgpu.ids > ${O}_bench.log; tail -c 300 ${O}_bench.log
