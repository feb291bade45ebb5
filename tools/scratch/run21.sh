#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for b in 16 32 64 128; do echo -n "B=$b "; python bench.py --batch $b --steps 40 --warmup 8 --cpu-seconds 0 --no-side-passes 2>&1 | grep "^{" | sed 's/.*"value": \([0-9.]*\).*"ms_per_step": \([0-9.]*\).*/\1 fps \2 ms/'; done
