#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for d in 0 1 2 4 8 3 5 6 10 12 14 15; do XFH_BX_DBG=$d timeout 100 python tools/bx_time.py 2>&1 | grep DBG; done
