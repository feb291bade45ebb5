#!/bin/bash
# Build the library as of git revision $1 (default HEAD) into gpurun_ab_old.so for tools/ab.sh.
REV=${1:-HEAD}
rm -rf /tmp/ab_old && mkdir -p /tmp/ab_old/a/csrc /tmp/ab_old/include
for f in $(git ls-tree --name-only $REV accelerated_features_amd/csrc/); do git show $REV:$f > /tmp/ab_old/a/csrc/$(basename $f); done
git show $REV:include/xfeat_hip.h > /tmp/ab_old/include/xfeat_hip.h
sed -i 's#"../../include/xfeat_hip.h"#"../../include/xfeat_hip.h"#' /tmp/ab_old/a/csrc/api.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -shared /tmp/ab_old/a/csrc/*.hip -o /root/repo/gpurun_ab_old.so && ls -la /root/repo/gpurun_ab_old.so
