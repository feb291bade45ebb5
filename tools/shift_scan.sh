#!/bin/bash
# Code-position scan of the WHOLE library under cold-start torture: libraries built with every matrix-core kernel moved by 4 N bytes
# (python -m accelerated_features_amd.build --shift N, N = 1 .. 15; build them BEFORE the GPU visit: tools/shift_scan.sh build), each soaked by tools/cold_soak.py.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
if [ "$1" == "build" ]; then for n in $(seq 1 15); do python -m accelerated_features_amd.build --shift $n > /dev/null 2>&1 || echo "build $n failed"; done; ls accelerated_features_amd/libxfeat_hip_shift*.so | wc -l; exit 0; fi
if [ "$1" == "kernels" ]; then      # the sensitive form: every matrix-core kernel alone in a tight cold-started loop, at every code position
  S=${SCAN_SECONDS:-3}
  : > gpurun_out/r04_shift_scan_kernels.txt
  timeout 200 python tools/conv_cold_scan.py $S 2>&1 | grep -v amdgpu.ids | grep "cold-started" >> gpurun_out/r04_shift_scan_kernels.txt
  for n in $(seq 1 15); do
    XFH_LIB_PATH=accelerated_features_amd/libxfeat_hip_shift$n.so timeout 200 python tools/conv_cold_scan.py $S 2>&1 | grep -v amdgpu.ids | grep "cold-started" >> gpurun_out/r04_shift_scan_kernels.txt
  done
  cat gpurun_out/r04_shift_scan_kernels.txt; exit 0
fi
S=${SCAN_SECONDS:-10}
OPTS=${SCAN_OPTS:-}      # e.g. SCAN_OPTS="heads_f32 0": the scan with the opt-in split-bf16 heads (the calibration of its sensitivity)
: > gpurun_out/r04_shift_scan.txt
timeout 120 python tools/cold_soak.py $S $OPTS 2>&1 | grep -v amdgpu.ids | grep cold_start >> gpurun_out/r04_shift_scan.txt
for n in $(seq 1 15); do
  XFH_LIB_PATH=accelerated_features_amd/libxfeat_hip_shift$n.so timeout 120 python tools/cold_soak.py $S $OPTS 2>&1 | grep -v amdgpu.ids | grep cold_start >> gpurun_out/r04_shift_scan.txt
done
cat gpurun_out/r04_shift_scan.txt
