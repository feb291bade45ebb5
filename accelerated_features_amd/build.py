"""Build libxfeat_hip.so (hipcc, gfx950 only) in-tree.

    python -m accelerated_features_amd.build [--force]

Sources: accelerated_features_amd/csrc/*.hip ; output: accelerated_features_amd/libxfeat_hip.so
(the .so is git-ignored; it travels to the GPU box with the repo snapshot).
"""
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libxfeat_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-ffp-contract=on"]


# per-file additions.  k_match_f16: without NaN semantics fmaxf(fmaxf(a, b), c) is ONE v_max3_f32 (else every operand is canonicalised
# first); descriptors are finite by construction.  No SLP: hipcc paired the refine's four fma chains into v_pk_fma_f32, shuffling the 64
# registers of the Y row into pairs first (32 more registers: spills).
EXTRA_FLAGS = {"k_match_f16.hip": ["-fno-honor-nans", "-fno-slp-vectorize"],
               # block1 is vector-issue bound (PMC): with NaN semantics every ReLU is a canonicalising v_max(x, x) + the v_max(x, 0); activations are finite
               "k_conv_direct.hip": ["-fno-honor-nans"],
               # conv_rs64 runs one wave per SIMD: every instruction beside its MFMA stream is an issue slot, and packed fp32 ops (v_pk_add_f32 / v_pk_mul_f32, which the SLP
               # vectoriser makes of the segment conversion's adjacent subtractions and multiplications) cost ~ 13 cycles each beside MFMAs instead of a slot
               "k_conv_rs64.hip": ["-fno-slp-vectorize", "-fno-honor-nans"]}      # (no NaN semantics: the range guard's max3 takes |x| operands without canonicalising them first)


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libxfeat_hip.so cannot be built (ROCm toolchain required)")


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    hdr = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdr.append(os.path.join(os.path.dirname(HERE), "include", "xfeat_hip.h"))
    return hdr


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, shift=0):
    """shift = N > 0: libxfeat_hip_shiftN.so -- EVERY matrix-core kernel's body moved by 4 N bytes (common.hpp: XFH_CODE_SHIFT; tools/bench_src/scan_probe.cpp)."""
    hipcc = _hipcc()
    tag = f"_shift{shift}" if shift else ""
    OBJ = globals()["OBJ"] + tag
    LIB = globals()["LIB"].replace(".so", tag + ".so")
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _deps()
    jobs = []
    for src in _sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ([f"-DXFH_CODE_SHIFT={shift}"] if shift else []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for src, rc, out in ex.map(compile_one, jobs):
                if verbose and out.strip():
                    print(out, file=sys.stderr)
                if rc != 0:
                    raise RuntimeError(f"hipcc failed on {src}:\n{out}")
                if verbose:
                    print("compiled", os.path.basename(src))
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in _sources()]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        if verbose:
            print("linked", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, shift=int(sys.argv[sys.argv.index("--shift") + 1]) if "--shift" in sys.argv else 0)
