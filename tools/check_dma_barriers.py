#!/usr/bin/env python3
"""ISA audit: in every kernel that uses the global->LDS DMA (global_load_lds* / buffer_load* ... lds), every
s_barrier must be directly preceded by an s_waitcnt with vmcnt(0) -- hipcc was observed to drop it (see common.hpp
lds_dma_barrier).  No partial counts once the kernel has issued a store: vmcnt orders loads only, stores are acknowledged out of order with respect to them
(k_conv_bx64s2.hip's first version left "the 16 youngest" -- its output stores -- in flight and read stale DMA data once in 1500 two-lane steps).
Exit code 1 and a listing on violation.   python tools/check_dma_barriers.py"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_asm  # noqa: E402


def audit(src):
    asm = isa_asm.asm(src)
    bad, n_kern, n_bar = [], 0, 0
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if "global_load_lds" not in body and not re.search(r"buffer_load_dword\w*[^\n]* lds", body):
            continue
        n_kern += 1
        ins, in_asm = [], False
        for l in body.splitlines():
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif in_asm and t.startswith("; xfh-no-dma-wave"):
                ins.append("asm:xfh-no-dma-wave")
            elif l.startswith("\t") and not t.startswith((";", ".")):
                ins.append(("asm:" if in_asm else "") + t)
        for i, l in enumerate(ins):
            if l.startswith("s_barrier"):
                n_bar += 1
                # walk back: a vmcnt(0) wait must come before any vector-memory instruction does
                ok, j = False, i - 1
                while j >= 0 and i - j < 192:      # (a kernel may zero its 64 accumulators between the wait and the barrier)
                    p = ins[j]
                    if "s_waitcnt" in p and "vmcnt(0)" in p:
                        ok = True
                        break
                    # conv_bx64s2x_kernel's staging waves: the marker says "this wave has issued no LDS-DMA" (the ring's DMA belongs to the other waves' copy of the
                    # code, which waits for it) -- checked here: no DMA instruction between this barrier and the wave's previous one
                    if p == "asm:xfh-no-dma-wave":
                        k = j - 1
                        while k >= 0 and not ins[k].startswith("s_barrier"):
                            k -= 1
                        ok = k >= 0 and not any(re.search(r"(global_load_lds|buffer_load_dword\w*[^\n]* lds)", q) for q in ins[k:j])
                        break
                    # a PARTIAL count is in order where the kernel has issued no store yet (loads return in order among themselves: the objection above is about
                    # stores) -- linear_fxd_kernel leaves the row pieces it has just requested in flight; its only stores are the epilogue's, behind the last such barrier
                    if p.startswith("asm:") and re.search(r"s_waitcnt vmcnt\(\d+\)", p) and not any(
                            q.replace("asm:", "").startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic")) for q in ins[:i]):
                        ok = True
                        break
                    # ... or where the wave has passed a vmcnt(0) since its last store: walking back from the partial wait, a full wait comes before any store does
                    # (conv_bx64s2x_kernel: row 0 of a unit waits for everything -- the previous unit's output stores -- rows 1 - 11 leave the youngest DMA in flight)
                    if p.startswith("asm:") and re.search(r"s_waitcnt vmcnt\(\d+\)", p):
                        k = j - 1
                        while k >= 0:
                            q = ins[k].replace("asm:", "")
                            if "s_waitcnt" in q and "vmcnt(0)" in q:
                                ok = True
                                break
                            if q.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic", "s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                                break
                            k -= 1
                        break
                    if p.replace("asm:", "").startswith(("global_", "buffer_", "flat_", "scratch_", "s_barrier", "s_cbranch", "s_branch")):
                        break
                    j -= 1
                if not ok:
                    bad.append((name, ins[max(0, i - 4):i]))
    return n_kern, n_bar, bad


def main():
    def _src_dma(f):      # the file and the kernel-body headers it includes (csrc/*_body.hpp), transitively
        t = open(f).read()
        return t + "".join(_src_dma(os.path.join(os.path.dirname(f), h)) for h in re.findall(r'#include "(\w+_body\.hpp)"', t))
    files = [f for f in sorted(glob.glob(os.path.join(ROOT, "accelerated_features_amd", "csrc", "*.hip")))
             if re.search(r"global_load_lds|buffer_load[^\n]* lds", _src_dma(f))]
    total_bad = 0
    for f in files:
        nk, nb, bad = audit(f)
        print(f"{os.path.basename(f)}: {nk} DMA kernels, {nb} barriers, {len(bad)} without vmcnt(0)")
        for name, prev in bad[:10]:
            print("   ", name[:90], "<-", " | ".join(prev))
        total_bad += len(bad)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
