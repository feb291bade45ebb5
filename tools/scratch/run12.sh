#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for V in "XFH_WINO_SKEW=0" "XFH_WINO_TUNE=2" "XFH_WINO_TUNE=2 XFH_WINO_SKEW=8,2" "XFH_WINO_TUNE=2 XFH_WINO_SKEW=14,2" "XFH_WINO_TUNE=2 XFH_WINO_SKEW=20,2" "XFH_WINO_SKEW=0"; do
    env $V python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-side-passes 2>&1 | V="$V" python -c "
import sys, json, os
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line)
        print('%-40s fps %9.1f  ms/step %.4f  wino24 %.1f us  convs %.1f us/step' % (os.environ['V'], d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline_conv_family']['us_per_step']))
"
done
