#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python tools/trace_block1.py 2>&1 | grep -v amdgpu
bash tools/scratch/run13.sh
