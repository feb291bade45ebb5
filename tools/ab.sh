#!/bin/bash
# In-run A/B of two builds of the library on the same GPU box (cross-run numbers vary by several %):
#   tools/ab.sh            # compares accelerated_features_amd/libxfeat_hip.so (new) with gpurun_ab_old.so (old)
# Build the "old" side first with tools/build_ab_old.sh <git-rev>.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do
  for L in "" "$PWD/gpurun_ab_old.so"; do
    XFH_LIB_PATH=$L python bench.py --steps 20 --warmup 5 --cpu-seconds 0 2>&1 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print('%-4s fps %9.1f  ms/step %.4f  match %.1f us  convs %.1f us/step' % ('${L:+old}' or 'new', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline_conv_family']['us_per_step']))
"
  done
done
