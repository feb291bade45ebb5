"""Print the headline of a bench.py run and the per-kernel rows whose name contains one of the given substrings:  python tools/show_detail.py <stdout log> <stderr log> [substring ...]"""
import json
import sys

for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print("value", d["value"], "ms_per_step", d["ms_per_step"], {k: d.get(k) for k in ("median_of_5x20_steps_fps", "single_lane_synchronous_fps", "public_api_fps") if k in d})
for l in open(sys.argv[2]):
    if l.startswith("# detail:"):
        d = json.loads(l[len("# detail:"):])
        for r in d.get("roofline_kernels", []):
            if len(sys.argv) <= 3 or any(k in r["kernel"] for k in sys.argv[3:]):
                print(f'  {r["kernel"][:70]:70s} {r["us"]:7.1f} us   floor {r["floor_us"]:6.1f}   frac {r["frac"]}')
