#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters: separate --pmc passes (FETCH_SIZE costs 3 of the
# 4 TCC slots, WRITE_SIZE 2), kernel-trace only, as MI355X_MICROARCH.md prescribes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $C -d "$OLDPWD/gpurun_out/pmc_$C" -o pmc --output-format csv -- python "$OLDPWD/bench.py" --steps 3 --warmup 2 --cpu-seconds 0 --no-side-passes --wake-ms 0 --lanes 1 > "$OLDPWD/gpurun_out/pmc_$C.log" 2>&1; echo "$C rc=$?")
done
python - <<'PY'
import csv, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(f"gpurun_out/pmc_{c}/pmc_counter_collection.csv")))
    agg = collections.defaultdict(list)
    for r in rows:
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("xfh::", "")].append(float(r["Counter_Value"]))
    out[c] = {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}
names = sorted(set(out["FETCH_SIZE"]) | set(out["WRITE_SIZE"]))
print("kernel | launches | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch")
table = {}
for n in names:
    f = out["FETCH_SIZE"].get(n, (0, 0)); w = out["WRITE_SIZE"].get(n, (0, 0))
    table[n] = {"launches": f[1], "fetch_kib": f[0], "write_kib": w[0]}
    print(f"{n[:60]:60s} {f[1]:4d} {f[0]:14.1f} {w[0]:14.1f}")
json.dump(table, open("gpurun_out/pmc_traffic_raw.json", "w"), indent=1)
PY
