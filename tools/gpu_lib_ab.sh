#!/bin/bash
# In-call A/B of two builds of libxfeat_hip.so under the default bench (boxes differ by +-5 %: only runs on ONE box compare).   gpurun -- 'bash tools/gpu_lib_ab.sh <a.so> <b.so> [rounds 2]'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; L=accelerated_features_amd/libxfeat_hip.so
cp $L /tmp/keep.so
for r in $(seq 1 ${3:-2}); do for so in "$1" "$2"; do
  cp "$so" $L
  python bench.py --cpu-seconds 0 --no-side-passes 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print('$so', j['value'], j['config']['wake_up_window_fps'][-3:])"
done; done
cp /tmp/keep.so $L
