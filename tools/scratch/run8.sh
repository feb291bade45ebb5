#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/bx_check.py > gpurun_out/bx_check.log 2>&1; echo "bx rc=$?"; tail -30 gpurun_out/bx_check.log | cut -c1-200
