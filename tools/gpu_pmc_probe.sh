#!/bin/bash
# PMC counters of one kernel under a stand-alone probe binary (counters only, two passes):  tools/gpu_pmc_probe.sh "<kernel-name-substring>[;<another>]" <probe command ...>
#   gpurun --timeout 600 -- 'bash tools/gpu_pmc_probe.sh mnn_f16_sweep gpurun_probe/tail_probe accelerated_features_amd/libxfeat_hip.so gpurun_probe/weights.bin 3 64 480 640 0'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out; rm -rf gpurun_out/pmcA gpurun_out/pmcB
K=$1; shift
CMD=""
for a in "$@"; do if [ -e "$R/$a" ]; then CMD="$CMD $R/$a"; else CMD="$CMD $a"; fi; done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d "$R/gpurun_out/pmcA" -o pmc --output-format csv -- $CMD > "$R/gpurun_out/pmcA.log" 2>&1; echo rcA=$?)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAVES -d "$R/gpurun_out/pmcB" -o pmc --output-format csv -- $CMD > "$R/gpurun_out/pmcB.log" 2>&1; echo rcB=$?)
python3 - "$K" <<'PY'
import csv, glob, sys, collections
for k in sys.argv[1].split(";"):
  print("==", k)
  for d in ("gpurun_out/pmcA", "gpurun_out/pmcB"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if k in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for name, v in sorted(acc.items()):
            print(f"  {name:28s} launches {len(v):3d}  mean per launch {sum(v)/len(v):16.0f}")
PY
