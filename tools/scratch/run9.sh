#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 --cpu-seconds 6 2>&1 | grep "^{" > gpurun_out/bench_r02.json; cut -c1-3000 gpurun_out/bench_r02.json
bash tools/gpu_traffic.sh > gpurun_out/traffic.log 2>&1; tail -40 gpurun_out/traffic.log
