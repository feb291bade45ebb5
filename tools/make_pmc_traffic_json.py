#!/usr/bin/env python3
"""gpurun_out/pmc_traffic_raw.json (tools/gpu_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes over bench.py) -> profiles/pmc_traffic.json:
HBM bytes per launch of every kernel of the step, the read side corrected as MI355X_MICROARCH.md prescribes (gfx950 counts 128-byte requests as 64 B for wide
coalesced reads: x 2, re-calibrated in the same run on gray_stats_kernel, which reads exactly B*3*H*W*4 bytes).      python tools/make_pmc_traffic_json.py <round tag>"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
raw = json.load(open(os.path.join(ROOT, "gpurun_out", "pmc_traffic_raw.json")))
B, H, W = 64, 480, 640
rgb = B * 3 * H * W * 4
gs = next((v for k, v in raw.items() if k.startswith("gray_stats_kernel")), None)
corr = 2.0
cal = None
if gs and gs["fetch_kib"] > 0:
    cal = rgb / (gs["fetch_kib"] * 1024.0)
table = {}
for k, v in raw.items():
    table[k] = {"launches": v["launches"], "hbm_read_bytes": int(v["fetch_kib"] * 1024 * corr), "hbm_write_bytes": int(v["write_kib"] * 1024),
                "hbm_bytes": int(v["fetch_kib"] * 1024 * corr + v["write_kib"] * 1024)}
b1 = next((k for k in table if k.startswith("block1_mx_kernel")), None) or next((k for k in table if k.startswith("block1_fused_kernel")), None)      # (the default block1 since round 5: block1_mx_kernel<7>)
out = {"collected": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 3 --warmup 2 --cpu-seconds 0 --no-side-passes --wake-ms 0 --lanes 1` (every launch at the bench shape), tools/gpu_traffic.sh, "
                    f"final build of the round ({tag}); raw table profiles/{tag}_pmc_traffic_raw.json; FETCH_SIZE doubled (gfx950 counts 128-B requests as 64 B): calibration in this run on "
                    f"gray_stats_kernel = {rgb} B of RGB per launch -> measured factor {cal:.3f}" if cal else "no calibration kernel found",
       "read_correction": corr, "read_calibration_on_gray_stats": cal,
       "dominant_kernel": b1.split("<")[0] if b1 else None,
       "dominant_kernel_hbm_bytes_per_launch": table[b1]["hbm_bytes"] if b1 else None,
       "dominant_kernel_algorithmic_bytes_per_launch": 4 * (B * H * W + 24 * B * H * W // 16),
       "kernels": table}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
json.dump(raw, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic_raw.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("dominant_kernel", "dominant_kernel_hbm_bytes_per_launch", "dominant_kernel_algorithmic_bytes_per_launch", "read_calibration_on_gray_stats")}))
for k, v in sorted(table.items(), key=lambda kv: -kv[1]["hbm_bytes"])[:16]:
    print(f"{k[:70]:70s} {v['launches']:4d}  read {v['hbm_read_bytes'] / 1e6:8.1f} MB  write {v['hbm_write_bytes'] / 1e6:8.1f} MB")
