#!/bin/bash
# Build the CURRENT working tree's kernels into gpurun_ab_old.so (the "old"/baseline side of tools/ab.sh).
# Workflow: build_variant.sh (baseline) -> edit kernels -> python -m accelerated_features_amd.build -> gpurun tools/ab.sh
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -fno-gpu-rdc -shared accelerated_features_amd/csrc/*.hip -o gpurun_ab_old.so && ls -la gpurun_ab_old.so
