#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for w in dense megadepth lighterglue; do
timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --cpu-seconds 5 2>&1 | grep "^{" > gpurun_out/bench_r02d_$w.json; cut -c1-260 gpurun_out/bench_r02d_$w.json
done
