#!/bin/bash
# The fine_matcher chain alone (tools/gpu_fine_time.py, 100 000 rows): per-kernel durations, HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes) and L2 hit / miss counts.
# Output: gpurun_out/pmc_fine_<tag>.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${1:-fine}
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1)); rm -rf gpurun_out/pmcf$i
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $SET -d "$OLDPWD/gpurun_out/pmcf$i" -o pmc --output-format csv -- python "$OLDPWD/tools/gpu_fine_time.py" 100000 > "$OLDPWD/gpurun_out/pmcf$i.log" 2>&1; echo rc$i=$?)
done
python3 - "$TAG" <<'PY'
import csv, glob, sys, collections
out = open(f"gpurun_out/pmc_fine_{sys.argv[1]}.txt", "w")
for d in sorted(glob.glob("gpurun_out/pmcf[0-9]")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "linear" in row["Kernel_Name"]:
                acc[(row["Kernel_Name"][:44], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, name), v in sorted(acc.items()):
            line = f"{k:44s} {name:24s} launches {len(v):3d}  mean per launch {sum(v)/len(v):16.0f}"
            print(line); out.write(line + "\n")
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[:1]:
        if d.endswith("1"):
            acc = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                if "linear" in row["Kernel_Name"]:
                    acc[row["Kernel_Name"][:44]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1000.0)
            for k, v in sorted(acc.items()):
                line = f"{k:44s} duration (under the counter pass) launches {len(v):3d}  mean {sum(v)/len(v):8.1f} us  min {min(v):8.1f}"
                print(line); out.write(line + "\n")
PY
rm -rf gpurun_out/pmcf[0-9]
