#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/trace_bx.py 2>&1 | grep -v amdgpu.ids | sed -n 4,12p
timeout 300 python tools/bx_check.py block2.0 > gpurun_out/bx_check.log 2>&1; echo "bx rc=$?"; grep -v amdgpu.ids gpurun_out/bx_check.log | tail -14 | cut -c1-200
