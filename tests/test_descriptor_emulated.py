"""descriptor_kernel (csrc/k_detect.hip) compiled for the HOST (tests/emu/) and run against a float64 restatement of xfeat.py:70,83-103: F.normalize(feats) -> bicubic sampling at the
selected key-points (the reference's fp32 coordinate arithmetic, zeros outside the map) -> F.normalize, the key-point / score / n_valid epilogue of the top-k and the fp16 copies
the matcher's filter reads.  The kernel's lane exchanges are DPP operands (row_newbcast of a tap's weight and byte offset, a butterfly for the norm): emu.hpp's update_dpp."""
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "accelerated_features_amd", "csrc")
EMU = os.path.join(ROOT, "tests", "emu")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _slice():
    t = open(os.path.join(CSRC, "k_detect.hip")).read()
    a = t.index("__device__ inline float sample_coord(int p, int S, int Sm) {")
    sc = t[a:t.index("\n}\n", a) + 3]
    a = t.index("__global__ __launch_bounds__(256) void descriptor_kernel(")
    k = t[a:t.index("void prof_begin(Profiler* p, int which, hipStream_t st);", a)]
    k = k.replace("__global__ __launch_bounds__(256) void descriptor_kernel(", "inline void descriptor_kernel(")
    s = (sc + "\n" + k).replace("__device__ ", "")
    assert "asm volatile" not in s and "__shared__" not in s and s.count("__builtin_amdgcn_update_dpp") == 3
    return s


@pytest.fixture(scope="module")
def emu_bin():
    if not os.path.exists(CLANG):
        pytest.skip("no host clang")
    td = tempfile.mkdtemp()
    open(os.path.join(td, "descriptor_slice.hpp"), "w").write(_slice())
    out = os.path.join(td, "descriptor_emu")
    subprocess.run([CLANG, "-O1", "-w", "-std=c++20", "-pthread", "-I", td, "-I", EMU, os.path.join(EMU, "descriptor_emu.cpp"), "-o", out], check=True)
    return out


def _float_ord(f):
    u = np.float32(f).view(np.uint32)
    return np.uint32(~u) if u & np.uint32(0x80000000) else np.uint32(u | np.uint32(0x80000000))


def _sample_coord(p, S, Sm):      # the kernel's (= the reference's) fp32 operation order; the last step is one fused multiply-add
    q = np.float32(p) / np.float32(S - 1)
    g1 = np.float32(np.float32(np.float32(2.0) * q) - np.float32(1.0)) + np.float32(1.0)
    return np.float32(np.float64(np.float32(g1)) * np.float64(np.float32(Sm) * np.float32(0.5)) - 0.5)


def _cubic(t):
    A = -0.75
    def near(x): return ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0
    def far(x): return ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A
    return [far(t + 1.0), near(t), near(1.0 - t), far(2.0 - t)]


@pytest.mark.parametrize("B,H,W,top_k,nsel", [(1, 64, 96, 64, [64]), (2, 96, 64, 48, [48, 21]), (1, 480, 640, 160, [150]), (9, 32, 32, 16, [16, 0, 3, 16, 7, 16, 1, 15, 16])])
def test_descriptor_kernel_on_the_host(emu_bin, B, H, W, top_k, nsel):
    rng = np.random.default_rng(B * 1000 + H + top_k)
    hc, wc, npix = H // 8, W // 8, (H // 8) * (W // 8)
    feats = (rng.standard_normal((B, npix, 64)) * rng.uniform(0.2, 3.0, (B, npix, 1))).astype(np.float32)
    inv = (np.float32(1.0) / np.maximum(np.sqrt((feats.astype(np.float32) ** 2).sum(-1, dtype=np.float32)), np.float32(1e-12))).astype(np.float32)
    cap = max(top_k, 200)
    cand = np.zeros((B, cap), np.uint32)
    skeys = np.full((B, top_k), np.uint64(0xffffffffffffffff), np.uint64)
    scores = np.zeros((B, top_k), np.float32)
    for b in range(B):
        # candidates in row-major order, the image's corners and borders among them (taps outside the map: zero weight, clamped address)
        forced = [(0, 0), (0, W - 1), (H - 1, 0), (H - 1, W - 1), (0, W // 2), (H // 2, 0), (H - 2, 3), (5, W - 3), (7, 7), (8, 8)]
        pix = sorted(set(forced) | {(int(y), int(x)) for y, x in zip(rng.integers(0, H, cap), rng.integers(0, W, cap))})[:cap]
        while len(pix) < cap:
            pix = sorted(set(pix) | {(int(rng.integers(0, H)), int(rng.integers(0, W)))})[:cap]
        cand[b] = [np.uint32((y << 16) | x) for y, x in pix]
        k = nsel[b]
        sc = np.sort(rng.uniform(0.01, 1.0, k).astype(np.float32))[::-1].copy()
        nz = max(0, k - int(rng.integers(0, 4)))          # a tail of zero scores: n_valid is the positive prefix
        sc[nz:] = 0.0
        slots = rng.permutation(cap)[:k]
        slots[nz:] = np.sort(slots[nz:])                  # equal scores: ascending slot (= ascending key)
        for j in range(k):
            skeys[b, j] = (np.uint64(np.uint32(~_float_ord(sc[j]))) << np.uint64(32)) | np.uint64(slots[j])
        scores[b, :k] = sc
        assert (np.diff(skeys[b, :k].astype(object)) > 0).all()
    rw, rh = np.float32(1.25), np.float32(0.75)
    blob = np.array([B, H, W, cap, top_k], np.int32).tobytes() + np.array([rw, rh], np.float32).tobytes() + feats.tobytes() + inv.tobytes() + cand.tobytes() + skeys.tobytes() + np.array(nsel, np.int32).tobytes()
    out = subprocess.run([emu_bin], input=blob, capture_output=True, check=True, timeout=900).stdout
    o = 0
    def take(n, dt):
        nonlocal o
        a = np.frombuffer(out[o:o + n * np.dtype(dt).itemsize], dt); o += a.nbytes
        return a
    kp = take(B * top_k * 2, np.float32).reshape(B, top_k, 2); sc_d = take(B * top_k, np.float32).reshape(B, top_k)
    de = take(B * top_k * 64, np.float32).reshape(B, top_k, 64); d16 = take(B * top_k * 64, np.float16).reshape(B, top_k, 64); nv = take(B, np.int32)
    assert o == len(out)
    fn = feats.astype(np.float64) * inv.astype(np.float64)[..., None]
    worst = 0.0
    for b in range(B):
        k = nsel[b]
        assert nv[b] == int((scores[b, :k] > 0).sum())
        assert np.array_equal(sc_d[b, :k], scores[b, :k]) and not sc_d[b, k:].any() and not kp[b, k:].any() and not de[b, k:].any() and not d16[b, k:].any()
        for j in range(k):
            c = int(cand[b, int(skeys[b, j] & np.uint64(0xffffffff))]); x, y = c & 0xffff, c >> 16
            assert kp[b, j, 0] == np.float32(x) * rw and kp[b, j, 1] == np.float32(y) * rh
            ux, uy = _sample_coord(x, W, wc), _sample_coord(y, H, hc)
            fx, fy = np.floor(ux), np.floor(uy)
            wx, wy = _cubic(float(np.float32(ux - fx))), _cubic(float(np.float32(uy - fy)))
            acc = np.zeros(64)
            for r in range(4):
                for i in range(4):
                    yy, xx = int(fy) - 1 + r, int(fx) - 1 + i
                    if 0 <= yy < hc and 0 <= xx < wc:
                        acc += wy[r] * wx[i] * fn[b, yy * wc + xx]
            ref = acc / max(np.sqrt((acc * acc).sum()), 1e-12)
            worst = max(worst, float(np.abs(de[b, j] - ref).max()))
    print(f"B {B} {H}x{W} top_k {top_k}: max |desc - float64| {worst:.3g}")
    assert worst <= 1.5e-6
    assert np.array_equal(d16, (de * np.float32(256.0)).astype(np.float16))          # the fp16 copies: RNE(256 * row) of the kernel's own fp32 rows
