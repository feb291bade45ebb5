#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db or *_kernel_stats.csv) into a
per-kernel table (calls, total, avg, min, max, share) -- the summary committed under profiles/.

    python tools/prof_summary.py gpurun_out/prof/r01_results.db > profiles/r01_kernel_stats.md
"""
import csv
import glob
import os
import sqlite3
import sys


def from_db(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    return list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                            "from kernels group by name order by 3 desc"))


def main():
    path = sys.argv[1]
    if os.path.isdir(path):
        path = max(glob.glob(os.path.join(path, "**", "*.db"), recursive=True), key=os.path.getmtime)
    rows = from_db(path)
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace summary of `{os.path.basename(path)}`\n")
    print(f"total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches"
          + (f"; {steps} bench steps -> {tot / 1e3 / steps:.1f} us of kernels per step" if steps else "") + "\n")
    print("| share | calls | total ms | avg us | min us | max us | kernel |")
    print("|---:|---:|---:|---:|---:|---:|---|")
    for name, n, s, a, mn, mx in rows:
        if s / tot < 0.0005:
            continue
        short = name.replace("xfh::", "").split("(")[0].replace("void ", "")
        print(f"| {100 * s / tot:.2f}% | {n} | {s / 1e6:.3f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | `{short}` |")


if __name__ == "__main__":
    main()
