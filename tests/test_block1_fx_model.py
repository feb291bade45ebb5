"""CPU model of block1's last convolution on the fp16 matrix cores (csrc/block1_fx.hpp, block1_mx_kernel): the LDS layouts of the fp16-pair
activations and of the compact weight image, the lane -> address maps and the K order of v_mfma_f32_16x16x32_f16, restated in numpy and checked against a
direct fp64 convolution; the host packer itself (the header compiled with g++) against the restatement, byte for byte."""
import os
import subprocess
import tempfile

import numpy as np

from test_fx_arithmetic import pair, weight_triple

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "accelerated_features_amd", "csrc")

C3H, C3W, NEVEN, ROWB = 17, 33, 17, 33 * 16
PLANE = C3H * ROWB
A_OFF, B_OFF, C_OFF, D_OFF, ZERO_OFF, W4_BYTES = 0, 6144, 6912, 10496, 10880, 11264


def lane_real(cb, s, lane):
    return (cb == 0 or (lane & 15) < 8) and (s < 2 or (lane >> 4) == 0)


def lane_off(cb, s, q, lane):
    """regions A .. D and Z of block1_fx.hpp, written out case by case"""
    ln, kg = lane & 15, lane >> 4
    if cb == 0 and s < 2:
        return A_OFF + 1024 * (3 * s + q) + 16 * lane
    if cb == 0:
        return (B_OFF + 48 * ln if kg == 0 else ZERO_OFF) + 16 * q
    if s < 2:
        return (C_OFF + 112 * (8 * kg + ln) if ln < 8 else ZERO_OFF) + 16 * (3 * s + q)
    return (D_OFF + 48 * ln if ln < 8 and kg == 0 else ZERO_OFF) + 16 * q


def pack_w4(w_kc):
    """w_kc[(ci * 9 + tap) * 24 + co] (fp32) -> the compact LDS image as fp16 bit patterns (uint16[W4_BYTES / 2])"""
    q0, q1, q2 = (v.astype(np.float16).view(np.uint16) for v in weight_triple(w_kc))
    out = np.zeros(W4_BYTES // 2, np.uint16)
    seen = np.zeros(W4_BYTES // 2, bool)
    for co in range(24):
        for t in range(9):
            cb, ln, s, kg = co >> 4, co & 15, t >> 2, t & 3
            lane = 16 * kg + ln
            assert lane_real(cb, s, lane)
            for k, q in enumerate((q0, q1, q2)):
                o = lane_off(cb, s, k, lane)
                assert o % 16 == 0 and o + 16 <= ZERO_OFF
                for j in range(8):
                    assert not seen[o // 2 + j]
                    seen[o // 2 + j] = True
                    out[o // 2 + j] = q[(j * 9 + t) * 24 + co]
    assert seen.sum() == 24 * 9 * 3 * 8 and not seen[ZERO_OFF // 2:].any()      # every real weight once, nothing in the zero block
    return out


def c3_pixel_off(r, c):
    return r * ROWB + 16 * (NEVEN + (c >> 1) if c & 1 else c >> 1)


def test_layout_offsets():
    # a lane without a real weight reads zeros: its addresses stay inside the zero block for every fragment of its record
    img = pack_w4(np.full(72 * 24, 0.30001, np.float32))                       # (every fragment of this weight is non-zero)
    for cb in range(2):
        for s in range(3):
            for q in range(3):
                offs = [lane_off(cb, s, q, lane) for lane in range(64)]
                assert all(o % 16 == 0 and 0 <= o and o + 16 <= W4_BYTES for o in offs)
                for lane, o in enumerate(offs):
                    zero = not img[o // 2:o // 2 + 8].any()
                    assert zero == (not lane_real(cb, s, lane)), (cb, s, q, lane)
                # distinct banks within every group of 16 lanes (a ds_read_b128 group), broadcast reads of the zero block aside
                for g in range(4):
                    real = [o for lane, o in enumerate(offs) if lane >> 4 == g and lane_real(cb, s, lane)]
                    banks = [(o // 4 + k) % 64 for o in real for k in range(4)]
                    assert len(set(banks)) == len(banks), (cb, s, q, g)
    offs = sorted(c3_pixel_off(r, c) for r in range(C3H) for c in range(C3W))
    assert offs == list(range(0, PLANE, 16))                                       # the parity layout is a permutation of the tile
    # the 16 lanes of a B-fragment group (one tap, output columns 0 .. 15) read 256 contiguous bytes
    for dx in range(3):
        o = [c3_pixel_off(4, 2 * ln + dx) for ln in range(16)]
        assert o == list(range(o[0], o[0] + 256, 16))


def test_host_packer_is_the_restatement():
    src = r'''
#include <cstdio>
#include <vector>
#include "weight_split.hpp"
#include "block1_fx.hpp"
int main() {
    std::vector<float> w(72 * 24);
    if (fread(w.data(), 4, w.size(), stdin) != w.size()) return 1;
    std::vector<uint16_t> out(xfh::b1fx::W4_BYTES / 2);
    xfh::b1fx::pack_w4(w.data(), out.data(), [](float v, uint16_t (&q)[3]) { xfh::split_weight(v, q); });
    fwrite(out.data(), 2, out.size(), stdout);
    return 0;
}
'''
    rs = np.random.RandomState(3)
    w = (rs.randn(72 * 24) * 0.4).astype(np.float32)
    w[:6] = [30.9, -1e-7, 6.1e-5, 0.0, -3.0e-5, 1.0]
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "p.cpp"), "w").write(src)
        subprocess.run(["g++", "-O1", "-std=c++17", "-I", CSRC, os.path.join(td, "p.cpp"), "-o", os.path.join(td, "p")], check=True)
        got = np.frombuffer(subprocess.run([os.path.join(td, "p")], input=w.tobytes(), capture_output=True, check=True).stdout, np.uint16)
    assert np.array_equal(got, pack_w4(w))


def mfma_16x16x32(a, b, acc):
    """a, b: (64 lanes, 8) fp32 values of the fp16 fragments; acc: (64 lanes, 4).  D[m][n] += sum_k A[m][k] B[k][n]; lane l holds A row l & 15 / B column l & 15,
    K values 8 (l >> 4) .. + 7, and D[4 (l >> 4) + j][l & 15]; exact products, the accumulator rounded to fp32 once per instruction."""
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for l in range(64):
        A[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = a[l]
        B[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = b[l]
    D = A @ B
    out = acc.copy()
    for l in range(64):
        for j in range(4):
            out[l, j] = np.float32(np.float64(acc[l, j]) + D[4 * (l >> 4) + j, l & 15])
    return out


def test_conv4_on_the_matrix_cores_is_the_convolution():
    rs = np.random.RandomState(4)
    c3 = np.maximum(rs.randn(C3H, C3W, 8), 0).astype(np.float32) * 2           # ReLU'd activations of conv3, tile coordinates
    c3[0, :, :] = 0                                                             # (a tile at the top border: the padding row is stored as zeros)
    w_kc = (rs.randn(72 * 24) * 0.3).astype(np.float32)
    bias = rs.randn(24).astype(np.float32)
    # reference: out[orow, ocol, co] = sum c3[2 orow + dy, 2 ocol + dx, ci] w[(ci * 9 + 3 dy + dx) * 24 + co]
    ref = np.zeros((8, 16, 24))
    for dy in range(3):
        for dx in range(3):
            for ci in range(8):
                ref += c3[dy:dy + 16:2, dx:dx + 32:2, ci, None].astype(np.float64) * w_kc[(ci * 9 + 3 * dy + dx) * 24:(ci * 9 + 3 * dy + dx) * 24 + 24].astype(np.float64)
    ref += bias
    # LDS images
    hi, lo = pair(c3)
    planes = np.zeros((2, PLANE // 2), np.uint16)
    for r in range(C3H):
        for c in range(C3W):
            o = c3_pixel_off(r, c) // 2
            planes[0, o:o + 8] = hi[r, c].astype(np.float16).view(np.uint16)
            planes[1, o:o + 8] = lo[r, c].astype(np.float16).view(np.uint16)
    w4 = pack_w4(w_kc)
    f16 = lambda u: u.view(np.float16).astype(np.float32)
    out = np.zeros((8, 16, 24), np.float32)
    for orow in range(8):                                                        # one wave per output row
        acc = np.zeros((2, 64, 4), np.float32)
        for s in range(3):
            xh = np.zeros((64, 8), np.float32); xl = np.zeros((64, 8), np.float32)
            for lane in range(64):
                ln, kg = lane & 15, lane >> 4
                t = min(4 * s + kg, 8)
                o = c3_pixel_off(2 * orow + t // 3, 2 * ln + t % 3) // 2
                xh[lane] = f16(planes[0, o:o + 8]); xl[lane] = f16(planes[1, o:o + 8])
            for cb in range(2):
                wq = np.zeros((3, 64, 8), np.float32)
                for q in range(3):
                    for lane in range(64):
                        o = lane_off(cb, s, q, lane) // 2
                        wq[q, lane] = f16(w4[o:o + 8])
                acc[cb] = mfma_16x16x32(wq[2], xh, acc[cb])
                acc[cb] = mfma_16x16x32(wq[1], xl, acc[cb])
                acc[cb] = mfma_16x16x32(wq[0], xh, acc[cb])
        for cb in range(2):
            for lane in range(64):
                ln, kg = lane & 15, lane >> 4
                for j in range(4):
                    co = 16 * cb + 4 * kg + j
                    if co < 24:
                        out[orow, ln, co] = np.float32(acc[cb, lane, j]) * np.float32(1.0 / 2048.0) + bias[co]
    err = np.abs(out - ref).max() / np.abs(ref).max()
    print(f"conv4 on 16x16x32 fp16-pair MFMAs vs fp64: max |err| / max |y| = {err:.3g}")
    assert err < 3e-7


# ---- conv3 (block1.2: 8 -> 8, stride 1) on the same instruction, a column of the product = a pair of adjacent pixels: block1_fused_kernel<7> -------------------
C2H, C2W, C2_NEVEN, C2_ROWB = 19, 35, 18, 35 * 16
C2_PLANE = C2H * C2_ROWB
NPAIR, W3_REC, W3_ZERO_OFF, W3_IMAGE = 17, 96, 24 * 96, 3072


def c2_pixel_off(r, c):
    return r * C2_ROWB + 16 * (C2_NEVEN + (c >> 1) if c & 1 else c >> 1)


def w3_lane_off(s, j, lane):
    ln, kg = lane & 15, lane >> 4
    dx = kg - (ln >> 3)                      # rows 8 .. 15 are the right pixel of the pair: its window starts one column later
    return (W3_REC * (8 * dx + (ln & 7)) if 0 <= dx <= 2 else W3_ZERO_OFF) + 16 * (2 * s + j)


def pack_w3(w_kc):
    """conv3's w_kc[(ci * 9 + tap) * 8 + co] -> records of q0 / q2 fragments per (dx, cout), K step = tap row (q1 is derived in the kernel)"""
    q0, _, q2 = (v.astype(np.float16).view(np.uint16) for v in weight_triple(w_kc))
    out = np.zeros(W3_IMAGE // 2, np.uint16)
    for co in range(8):
        for dy in range(3):
            for dx in range(3):
                for j, q in enumerate((q0, q2)):
                    o = (W3_REC * (8 * dx + co) + 16 * (2 * dy + j)) // 2
                    assert not out[o:o + 8].any()
                    out[o:o + 8] = q[(np.arange(8) * 9 + 3 * dy + dx) * 8 + co]
    assert not out[W3_ZERO_OFF // 2:].any()
    return out


def test_conv3_host_packer_is_the_restatement():
    src = r"""
#include <cstdio>
#include <vector>
#include "weight_split.hpp"
#include "block1_fx.hpp"
int main() {
    std::vector<float> w(72 * 8);
    if (fread(w.data(), 4, w.size(), stdin) != w.size()) return 1;
    std::vector<uint16_t> out(xfh::b1fx::W3_IMAGE_BYTES / 2);
    xfh::b1fx::pack_w3(w.data(), out.data(), [](float v, uint16_t (&q)[3]) { xfh::split_weight(v, q); });
    fwrite(out.data(), 2, out.size(), stdout);
    return 0;
}
"""
    rs = np.random.RandomState(5)
    w = (rs.randn(72 * 8) * 0.4 + 0.01).astype(np.float32)
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "p.cpp"), "w").write(src)
        subprocess.run(["g++", "-O1", "-std=c++17", "-I", CSRC, os.path.join(td, "p.cpp"), "-o", os.path.join(td, "p")], check=True)
        got = np.frombuffer(subprocess.run([os.path.join(td, "p")], input=w.tobytes(), capture_output=True, check=True).stdout, np.uint16)
    assert np.array_equal(got, pack_w3(w))


def test_conv3_on_the_matrix_cores_is_the_convolution():
    rs = np.random.RandomState(6)
    c2 = np.maximum(rs.randn(C2H, C2W, 8), 0).astype(np.float32) * 2
    w_kc = (rs.randn(72 * 8) * 0.3).astype(np.float32)
    w_kc[:4] = [1e-5, -2e-5, 6.2e-5, 3e-6]                                       # weights whose fp16 image is subnormal: the derived q1 may differ from fp16(w) in its last bit
    bias = rs.randn(8).astype(np.float32)
    ref = np.zeros((C3H, C3W, 8))
    for dy in range(3):
        for dx in range(3):
            for ci in range(8):
                k = (ci * 9 + 3 * dy + dx) * 8
                ref += c2[dy:dy + C3H, dx:dx + C3W, ci, None].astype(np.float64) * w_kc[k:k + 8].astype(np.float64)
    ref = np.maximum(ref + bias, 0)
    hi, lo = pair(c2)
    # what lies behind the planes is NOT fp16-clean in the kernel (other tiles, uninitialised LDS): NaN patterns there must never be read
    planes = np.full((2, C2_PLANE // 2 + 64), 0x7e00, np.uint16)
    for r in range(C2H):
        for c in range(C2W):
            o = c2_pixel_off(r, c) // 2
            planes[0, o:o + 8] = hi[r, c].astype(np.float16).view(np.uint16)
            planes[1, o:o + 8] = lo[r, c].astype(np.float16).view(np.uint16)
    assert sorted(c2_pixel_off(r, c) for r in range(C2H) for c in range(C2W)) == list(range(0, C2_PLANE, 16))
    w3 = pack_w3(w_kc)
    f16 = lambda u: u.view(np.float16).astype(np.float32)
    out = np.full((C3H, C3W, 8), np.nan, np.float32)
    nblk = (C3H * NPAIR + 15) // 16
    assert nblk == 19
    for blk in range(nblk):
        acc = np.zeros((64, 4), np.float32)
        for s in range(3):
            xh = np.zeros((64, 8), np.float32); xl = np.zeros((64, 8), np.float32); a0 = np.zeros((64, 8), np.float32); a2 = np.zeros((64, 8), np.float32)
            for lane in range(64):
                ln, kg = lane & 15, lane >> 4
                ep = min(blk * 16 + ln, C3H * NPAIR - 1)
                r, pc = divmod(ep, NPAIR)
                col = 2 * pc + kg
                if col > C2W - 1:
                    col -= 2
                o = c2_pixel_off(r + s, col) // 2
                xh[lane] = f16(planes[0, o:o + 8]); xl[lane] = f16(planes[1, o:o + 8])
                o0, o2 = w3_lane_off(s, 0, lane) // 2, w3_lane_off(s, 1, lane) // 2
                a0[lane] = f16(w3[o0:o0 + 8]); a2[lane] = f16(w3[o2:o2 + 8])
            assert np.isfinite(xh).all() and np.isfinite(xl).all()
            a1 = (a0 * np.float32(2.0 ** -11)).astype(np.float16).astype(np.float32)          # v_pk_mul_f16 by 2^-11
            acc = mfma_16x16x32(a2, xh, acc)
            acc = mfma_16x16x32(a1, xl, acc)
            acc = mfma_16x16x32(a0, xh, acc)
        for lane in range(64):                                                    # lane (pair ln, kg): couts 4 (kg & 1) + j of pixel 2 pc + (kg >> 1)
            ln, kg = lane & 15, lane >> 4
            ep = blk * 16 + ln
            if ep < C3H * NPAIR:
                r, pc = divmod(ep, NPAIR)
                c = 2 * pc + (kg >> 1)
                if c < C3W:
                    for j in range(4):
                        co = 4 * (kg & 1) + j
                        assert np.isnan(out[r, c, co])                            # every output exactly once
                        out[r, c, co] = max(np.float32(acc[lane, j]) * np.float32(1.0 / 2048.0) + bias[co], 0)
    assert not np.isnan(out).any()
    err = np.abs(out - ref).max() / np.abs(ref).max()
    print(f"conv3 (pixel pairs) on 16x16x32 fp16-pair MFMAs (q1 derived from q0) vs fp64: max |err| / max |y| = {err:.3g}")
    assert err < 3e-7
