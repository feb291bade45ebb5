#!/usr/bin/env python3
"""Size histogram of the MegaDepth-1500 pair list (BASELINE configs[3]) -> accelerated_features_amd/data/megadepth1500_sizes.json.

Reads /root/reference/assets/megadepth_1500.json (1500 pairs; only `size0_hw` / `size1_hw` are used -- the images
themselves are not in the repository) and writes [[h0, w0, h1, w1, count], ...] sorted by count.  bench.py
--workload megadepth expands it to 1500 synthetic pairs at long side 1600.
    python tests/golden/make_megadepth_sizes.py"""
import collections
import json
import os

SRC = "/root/reference/assets/megadepth_1500.json"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "accelerated_features_amd", "data", "megadepth1500_sizes.json")

if __name__ == "__main__":
    d = json.load(open(SRC))
    c = collections.Counter((tuple(e["size0_hw"]), tuple(e["size1_hw"])) for e in d)
    rows = sorted(([a[0], a[1], b[0], b[1], n] for (a, b), n in c.items()), key=lambda r: (-r[4], r[:4]))
    assert sum(r[4] for r in rows) == len(d) == 1500
    json.dump(rows, open(DST, "w"), separators=(",", ":"))
    print(f"{len(rows)} distinct size pairs, {len(d)} pairs -> {DST}")
