#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python tools/trace_bx64.py 2>&1 | grep -v amdgpu.ids | tail -17
timeout 300 python tools/bx_check.py block_fusion.0 > gpurun_out/bx_check64.log 2>&1; echo "bx rc=$?"; grep -v amdgpu.ids gpurun_out/bx_check64.log | tail -14 | cut -c1-200
