#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table from hipcc -Rpass-analysis=kernel-resource-usage.
    python tools/kernel_resources.py accelerated_features_amd/csrc/k_conv_mfma.hip [...]
"""
import re
import subprocess
import sys

for src in sys.argv[1:]:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=on",
                        "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"], capture_output=True, text=True)
    cur = {}
    rows = []
    for line in r.stderr.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            if cur:
                rows.append(cur)
            cur = {"name": t.split(":", 1)[1].strip()}
        elif ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    if cur:
        rows.append(cur)
    print(f"== {src}")
    for c in rows:
        name = subprocess.run(["c++filt", c["name"]], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void xfh::", "")
        print(f"{name[:70]:70s} VGPR {c.get('VGPRs','?'):>4} AGPR {c.get('AGPRs','?'):>3} SGPR {c.get('SGPRs','?'):>3} "
              f"scratch {c.get('ScratchSize [bytes/lane]','?'):>4} occ {c.get('Occupancy [waves/SIMD]','?')} "
              f"spill s/v {c.get('SGPRs Spill','?')}/{c.get('VGPRs Spill','?')} LDS {c.get('LDS Size [bytes/block]','?')}")
