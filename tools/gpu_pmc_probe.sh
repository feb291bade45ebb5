#!/bin/bash
# SQ counters of ONE kernel launched by the stand-alone probe (no Python: ~30 s of GPU budget per counter set):  tools/gpu_pmc_probe.sh <case> <variant> [tag]
# counters only (no sys / hip traces).  Output: gpurun_out/pmc_probe_<tag>.txt (mean per launch of every counter, per kernel name)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
C=${1:-2}; V=${2:-12}; TAG=${3:-rs64}
L=$PWD/accelerated_features_amd/libxfeat_hip.so; W=$PWD/gpurun_probe/weights.bin; P=$PWD/gpurun_probe/rs64_probe
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf gpurun_out/pmcp$i
  (cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $SET -d "$OLDPWD/gpurun_out/pmcp$i" -o pmc --output-format csv -- $P $L $W one $C $V 5 > "$OLDPWD/gpurun_out/pmcp$i.log" 2>&1; echo rc$i=$?)
done
python3 - "$TAG" <<'PY'
import csv, glob, sys, collections
out = open(f"gpurun_out/pmc_probe_{sys.argv[1]}.txt", "w")
for d in sorted(glob.glob("gpurun_out/pmcp[0-9]")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            acc[(row["Kernel_Name"][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, name), v in sorted(acc.items()):
            line = f"{k:60s} {name:34s} launches {len(v):3d}  mean per launch {sum(v)/len(v):16.0f}"
            print(line); out.write(line + "\n")
PY
rm -rf gpurun_out/pmcp[0-9]
