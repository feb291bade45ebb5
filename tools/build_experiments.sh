#!/bin/bash
# experiment builds of the library: gpurun_exp<N>.so = -DXFH_EXPERIMENT=N (compile-time hooks in the kernels; measurements only)
cd /root/repo
for N in "$@"; do
  mkdir -p /tmp/exp$N
  for f in accelerated_features_amd/csrc/*.hip; do
    b=$(basename $f .hip)
    extra=""; [ "$b" == "k_match_f16" ] && extra="-fno-honor-nans -fno-slp-vectorize"
    if grep -q XFH_EXPERIMENT $f; then
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=on $extra -DXFH_EXPERIMENT=$N -c $f -o /tmp/exp$N/$b.o &
    else
      cp accelerated_features_amd/_obj/$b.o /tmp/exp$N/$b.o
    fi
  done
  wait
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o gpurun_exp$N.so /tmp/exp$N/*.o && ls -la gpurun_exp$N.so
done
