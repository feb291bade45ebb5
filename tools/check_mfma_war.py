#!/usr/bin/env python3
"""ISA audit: no VALU instruction may write a register that one of the few preceding bf16 MFMAs reads as its A or B operand.

The K = 16 bf16 MFMAs of gfx950 read 16 bytes per lane and operand; a VALU result written a few cycles after the MFMA was issued landed
in operand lanes the matrix core had not read yet (split-bf16 heads: the cells of lanes 16-31 of a wave wrong in a few launches out of
many).  hipcc's hazard recogniser covers SrcC only, so the kernels keep a just-read fragment alive (or idle) for a while; this audit
checks the generated code.      python tools/check_mfma_war.py"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_asm  # noqa: E402
DIST = 8          # issue slots: every instruction counts one, s_nop N counts N + 1, an MFMA counts 8 (the pipe takes one at a time, 8 passes each)
sys.path.insert(0, ROOT)


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def audit(src):
    asm = isa_asm.asm(src)
    n_kern, n_mfma, bad = 0, 0, []
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if not re.search(r"v_mfma_f32_\d+x\d+x\d+_(bf16|f16)", body):
            continue
        n_kern += 1
        recent, n = [], 0
        for l in body.splitlines():
            t = l.strip().split()
            if not t or not l.startswith("\t") or t[0].startswith((";", ".")):
                continue
            args = [a.strip(",") for a in t[1:]]
            if t[0].startswith("v_mfma") and ("bf16" in t[0] or t[0].endswith("_f16")):
                recent = (recent + [(n, _regs(args[1]) | _regs(args[2]))])[-6:]
                n_mfma += 1
                n += 7
            elif t[0] == "s_nop" and args:
                n += int(args[0], 0)
            elif t[0].startswith("v_") and args:
                w = _regs(args[0])
                for k, r in recent:
                    if n - k <= DIST and (w & r):
                        bad.append((name, l.strip(), n - k))
            n += 1
    return n_kern, n_mfma, bad


def main():
    def _src(f):      # the file and the kernel-body headers it includes (csrc/*_body.hpp), transitively
        t = open(f).read()
        return t + "".join(_src(os.path.join(os.path.dirname(f), h)) for h in re.findall(r'#include "(\w+_body\.hpp)"', t))
    files = [f for f in sorted(glob.glob(os.path.join(ROOT, "accelerated_features_amd", "csrc", "*.hip"))) if re.search(r"mfma_f32_\d+x\d+x\d+_(bf16|f16)\(", _src(f))]
    total = 0
    for f in files:
        nk, nm, bad = audit(f)
        print(f"{os.path.basename(f)}: {nk} kernels, {nm} bf16 MFMAs, {len(bad)} operand registers rewritten within {DIST} instructions")
        for name, ins, d in bad[:10]:
            print("   ", name[:80], "|", ins, "|", d, "instructions after the MFMA")
        total += len(bad)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
