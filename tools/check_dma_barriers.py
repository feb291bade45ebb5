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
        # conv_bx64s2x_kernel: partial counts by design (vmcnt(3) leaves the weight row requested last in flight; conv_bx64s2_body.hpp argues why that is sound beside the
        # wave's stores).  What its waits guarantee is checked by RUNNING the same source with the DMA delivered as late as they allow (tests/emu/emu.hpp EMU_DEFER_DMA,
        # tests/test_conv_bx64s2_emulated.py), not by this structural lint; the kernel says so with a marker in its code
        if "; xfh-dma-protocol-emulated" in body:
            n_bar += len(re.findall(r"^\ts_barrier", body, re.M))
            continue
        ins, in_asm = [], False
        for l in body.splitlines():
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
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
                    # a PARTIAL count is in order where the kernel has issued no store yet (loads return in order among themselves: the objection above is about
                    # stores) -- linear_fxd_kernel leaves the row pieces it has just requested in flight; its only stores are the epilogue's, behind the last such barrier
                    if p.startswith("asm:") and re.search(r"s_waitcnt vmcnt\(\d+\)", p) and not any(
                            q.replace("asm:", "").startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic")) for q in ins[:i]):
                        ok = True
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
