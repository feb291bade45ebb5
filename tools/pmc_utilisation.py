"""Per-kernel effective clock and pipe utilisation from one rocprofv3 --pmc pass (SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU ... + --kernel-trace):
   python tools/pmc_utilisation.py <dir with pmc_kernel_trace.csv and pmc_counter_collection.csv>
SQ_BUSY_CYCLES is summed over the 32 shader engines of the chip (8 XCDs x 4): / 32 = the kernel's length in shader clocks, / its duration = the clock it ran at;
SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs = matrix-pipe cycles per SIMD (32 per v_mfma_f32_32x32x16_f16); SQ_ACTIVE_INST_VALU is counted in quad-cycles."""
import collections
import csv
import sys

d = sys.argv[1]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(d + "/pmc_kernel_trace.csv")):
    dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(d + "/pmc_counter_collection.csv")):
    cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
med = lambda x: sorted(x)[len(x) // 2]
rows = []
for k, v in dur.items():
    c = cnt[k]
    if "SQ_BUSY_CYCLES" not in c:
        continue
    us, busy = med(v), med(c["SQ_BUSY_CYCLES"]) / 32
    rows.append((us, k, busy, med(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024, med(c["SQ_ACTIVE_INST_VALU"]) * 4 / 1024, len(v)))
print(f"{'kernel':62s} {'n':>4s} {'us':>7s} {'kcycles':>8s} {'GHz':>5s} {'mfma %':>7s} {'valu %':>7s}")
for us, k, busy, mf, va, n in sorted(rows, reverse=True):
    if us < 1.0:
        continue
    name = k.replace("void ", "").replace("xfh::", "")
    name = name[: name.index("(")] if "(" in name and not name.startswith("_Z") else name
    print(f"{name[:62]:62s} {n:4d} {us:7.1f} {busy / 1e3:8.1f} {busy / us / 1e3:5.2f} {100 * mf / busy:7.1f} {100 * va / busy:7.1f}")
