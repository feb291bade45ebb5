cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; rm -rf gpurun_out/pmcd
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d "$OLDPWD/gpurun_out/pmcd" -o pmc --output-format csv -- python "$OLDPWD/bench.py" --workload dense --steps 1 --warmup 1 --cpu-seconds 0 > "$OLDPWD/gpurun_out/pmcd.log" 2>&1; echo rc=$?)
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/pmcd/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"]
        if "linear_fx" in k or "resize2" in k:
            acc[(k[:48], row["Counter_Name"])].append(float(row["Counter_Value"]))
    out=open("gpurun_out/r05_pmc_dense_linear_fx.txt","w")
    for (k, name), v in sorted(acc.items()):
        line=f"{k:48s} {name:28s} launches {len(v):3d}  mean per launch {sum(v)/len(v):16.0f}"
        print(line); out.write(line+"\n")
PY
rm -rf gpurun_out/pmcd
