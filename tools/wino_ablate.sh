#!/bin/bash
# Build ablation variants of the Winograd kernel (pieces of the main loop compiled out; results are
# garbage, only the timing matters):  tools/wino_ablate.sh 1 2 4 8 ...   -> gpurun_ab_w<N>.so
cd /root/repo
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -ffp-contract=on"
for n in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DWINO_SKIP=$n -c accelerated_features_amd/csrc/k_conv_wino.hip -o /tmp/k_conv_wino_$n.o || exit 1
  objs=$(ls accelerated_features_amd/_obj/*.o | grep -v k_conv_wino.o)
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o gpurun_ab_w$n.so $objs /tmp/k_conv_wino_$n.o || exit 1
  echo built gpurun_ab_w$n.so
done
