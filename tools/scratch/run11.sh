#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "winograd_configurations or repeated_launches" 2>&1 | tail -3
bash tools/ab_env.sh XFH_WINO_TUNE=4 2>&1 | grep fps
