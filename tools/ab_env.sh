#!/bin/bash
# In-run A/B of one library under two environments (cross-run numbers vary by several %):
#   tools/ab_env.sh XFH_SIDE=1            # "new" = with the variable set, "old" = without
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2 3; do
  for V in "$1" "${2:-XFH_AB_DUMMY=0}"; do
    env $V python bench.py --steps 20 --warmup 5 --cpu-seconds 0 2>&1 | V="$V" python -c "
import sys, json, os
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print('%-22s fps %9.1f  ms/step %.4f  match %.1f us  convs %.1f us/step' % (os.environ['V'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline_conv_family']['us_per_step']))
"
  done
done
