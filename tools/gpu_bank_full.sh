#!/bin/bash
# The round's closing visit: tools/gpu_bank.sh <tag> (suite, smoke, bench line, rocprofv3 stats with two lanes and one), then the HBM traffic of every kernel of the
# shipped binary (tools/gpu_traffic.sh: separate --pmc passes), the per-kernel clock / pipe-utilisation table (tools/gpu_pmc_kernel.sh + tools/pmc_utilisation.py), the
# per-kernel span table of the stand-alone probe, and one line each of the other workloads.  ~12 GPU-min.
#   gpurun --timeout 3000 -- 'bash tools/gpu_bank_full.sh r06_final'
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${1:-bank}; O=gpurun_out/$T
sha256sum accelerated_features_amd/libxfeat_hip.so | cut -c1-16 > ${O}_lib_sha16.txt
bash tools/gpu_bank.sh $T
bash tools/gpu_traffic.sh > ${O}_traffic.log 2>&1; tail -3 ${O}_traffic.log
bash tools/gpu_pmc_kernel.sh "block1_mx" "bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-side-passes --lanes 1" > ${O}_pmc_block1.txt 2>&1
python tools/pmc_utilisation.py gpurun_out/pmcA > ${O}_utilisation.txt 2>&1; head -30 ${O}_utilisation.txt
timeout 120 gpurun_probe/tail_probe accelerated_features_amd/libxfeat_hip.so gpurun_probe/weights.bin 20 > ${O}_tail_probe.txt 2>&1
for w in dense megadepth lighterglue; do timeout 600 python bench.py --workload $w 2>/dev/null | grep "^{" > ${O}_bench_$w.json; tail -c 200 ${O}_bench_$w.json; echo; done
