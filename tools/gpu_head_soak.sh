#!/bin/bash
# One GPU visit of the focused key-point-head soak (tools/head_soak.py); logs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
S=${SOAK_SECONDS:-45}
TAG=${SOAK_TAG:-head_soak}
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/bench_src/f16_denorm.hip -o /tmp/f16_denorm 2>/dev/null && /tmp/f16_denorm | tee gpurun_out/f16_denorm.log
timeout 900 python tools/head_soak.py --variants "${SOAK_VARIANTS:-0}" --foreign "${SOAK_FOREIGN:-backbone}" --iters 100000000 --max-seconds $S 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$TAG.log
if [ -n "$SOAK_VARIANTS2" ]; then
timeout 900 python tools/head_soak.py --variants "$SOAK_VARIANTS2" --foreign "${SOAK_FOREIGN2:-none}" --iters 100000000 --max-seconds $S 2>&1 | grep -v amdgpu.ids | tee gpurun_out/${TAG}_2.log
fi
