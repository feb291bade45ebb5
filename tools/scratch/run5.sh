#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for F in "" "-DB1_CIW=0" "-DB1_CIL=0" "-DB1_CIW=0 -DB1_CIL=0"; do
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on $F -Wno-unused-result -Iinclude tools/bench_src/block1_bench.hip -o /tmp/block1_bench 2>/dev/null && echo "== flags: $F" && /tmp/block1_bench 2>&1 | grep "variant 1"
done
