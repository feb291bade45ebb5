#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
XFH_BX=0 timeout 300 python tools/bx_check.py block3.0 > gpurun_out/bx_check_s2.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids gpurun_out/bx_check_s2.log | tail -16 | cut -c1-200
