#!/bin/bash
# PMC counters for the kernels of one bench step (counters only: no sys/hip traces):  tools/gpu_pmc_kernel.sh <kernel-name-substring>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/pmcA gpurun_out/pmcB
K=${1:-block1_fused}
# optional 2nd argument: the command to profile (default: two steps of the sparse bench)
CMD=${2:-"bench.py --steps 2 --warmup 1 --cpu-seconds 0"}
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d "$OLDPWD/gpurun_out/pmcA" -o pmc --output-format csv -- python $(echo "$CMD" | sed "s#^#$OLDPWD/#") > "$OLDPWD/gpurun_out/pmcA.log" 2>&1; echo rcA=$?)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAVES -d "$OLDPWD/gpurun_out/pmcB" -o pmc --output-format csv -- python $(echo "$CMD" | sed "s#^#$OLDPWD/#") > "$OLDPWD/gpurun_out/pmcB.log" 2>&1; echo rcB=$?)
python - "$K" <<'PY'
import csv, glob, sys, collections
ks = sys.argv[1].split(";")
for k in ks:
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
