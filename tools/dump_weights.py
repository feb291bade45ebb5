#!/usr/bin/env python3
"""The synthetic test weights (tests/fixtures.py) as the flat file the stand-alone probes read (tools/bench_src/r5_probe.cpp, rs64_probe.cpp): int32 count, then per array
int32 n + n fp32 -- the arrays of XFeatModel.weight_arrays() in the order of xfh_create.  Runs on CPU (no library needed).
    python tools/dump_weights.py gpurun_probe/weights.bin"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(path):
    import fixtures
    from accelerated_features_amd.spec import CONVS, FINE
    sd = fixtures.synthetic_state_dict()
    sd = {k[4:] if k.startswith("net.") else k: v for k, v in sd.items()}
    out = []
    for c in CONVS:
        keys = [f"{c.name}.layer.0.weight", f"{c.name}.layer.1.running_mean", f"{c.name}.layer.1.running_var"] if c.kind == "bn" else [f"{c.name}.weight", f"{c.name}.bias"]
        out += [sd[k] for k in keys]
    for li, fin, fout, bi in FINE:
        out += [sd[f"fine_matcher.{li}.weight"], sd[f"fine_matcher.{li}.bias"]]
        if bi is not None:
            out += [sd[f"fine_matcher.{bi}.running_mean"], sd[f"fine_matcher.{bi}.running_var"]]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(np.int32(len(out)).tobytes())
        for t in out:
            a = np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy()).reshape(-1)
            f.write(np.int32(a.size).tobytes())
            f.write(a.tobytes())
    print(f"{path}: {len(out)} arrays, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_probe/weights.bin")
