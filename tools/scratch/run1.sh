#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 -Wno-unused-result tools/bench_src/pk_fma_rate.hip -o /tmp/pk_fma_rate && /tmp/pk_fma_rate 2>&1 | tee gpurun_out/pk_fma_rate.log
python tools/scratch/debug_bm.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/debug_bm.log
