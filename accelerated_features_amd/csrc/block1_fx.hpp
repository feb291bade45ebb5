// block1's last convolution (block1.3: 8 -> 24, 3x3, stride 2; modules/model.py:40-48) on the fp16 matrix cores in the fp16-pair arithmetic
// (bx_split.hpp): layouts shared by the host packer (api.hip) and block1_fused_kernel<6> (k_conv_direct.hip), restated in numpy by
// tests/test_block1_fx_model.py.
//
// Instruction: v_mfma_f32_16x16x32_f16, D[cout][pixel] += A[cout][k] B[k][pixel].  Lane l = (ln = l & 15, kg = l >> 4):
//   A (weights): row = cout 16 cb + ln, K values 8 kg .. 8 kg + 7      B (pixels): column = output column ln of the wave's output row, the same K values
//   D: lane (pixel ln, kg) holds couts 16 cb + 4 kg + j, j = 0 .. 3
// K = 9 taps x 8 channels = 72, padded to 96 = three K steps of four taps: K value 8 kg + j of step s = channel j of tap t = 4 s + kg
// (t = 3 dy + dx; taps 9 .. 11 have zero weights and re-read tap 8).  Two cout blocks (24 couts of 32), three products per step and block:
// (q2, xh) (q1, xl) (q0, xh) -- 18 MFMAs per wave and tile, one wave per output row of the 8 x 16 tile.
//
// Weights in LDS, compact: only the lanes that hold a real (cout, tap) are stored; every other lane reads zeros.  q = 0, 1, 2 -> q0, q1, q2
// (weight_split.hpp: split_weight mode 1), 16 bytes (8 fp16: channels 0 .. 7) per lane and fragment.
//   A  [    0,  6144)  cb 0, s < 2 (every lane real): fragment-major, (3 s + q) KiB + 16 lane
//   B  [ 6144,  6912)  cb 0, s = 2 (tap 8: lanes kg == 0): lane-major, 48 ln + 16 q
//   C  [ 6912, 10496)  cb 1, s < 2 (couts 16 .. 23: lanes ln < 8): lane-major, 112 (8 kg + ln) + 16 (3 s + q)   (112 = 7 x 16: the eight lanes of a group on distinct banks)
//   D  [10496, 10880)  cb 1, s = 2: lane-major, 48 ln + 16 q
//   Z  [10880, 11264)  zeros: a lane without a real weight reads Z + the same fragment offset its record would have
// (lane-major records: ONE base address per lane and region, the fragment in the instruction's offset field -- whether a lane is real is decided once, not per fragment)
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define XFH_HD __host__ __device__
#else
#define XFH_HD
#endif

namespace xfh {
namespace b1fx {

constexpr int NCO = 24, NCI = 8, NTAP = 9, NSTEP = 3;
constexpr int A_OFF = 0, B_OFF = 6144, C_OFF = 6912, D_OFF = 10496, ZERO_OFF = 10880;
constexpr int B_REC = 48, C_REC = 112, D_REC = 48;
constexpr int W4_BYTES = 11 * 1024;                                   // eleven 1-KiB pieces of LDS-DMA
static_assert(B_OFF == 6 * 1024 && C_OFF == B_OFF + 16 * B_REC && D_OFF == C_OFF + 32 * C_REC && ZERO_OFF == D_OFF + 8 * D_REC && ZERO_OFF + C_REC <= W4_BYTES, "layout");

// per-lane base of a region (the kernel computes these four once) and the fragment's offset within the lane's record
XFH_HD inline int lane_base(int cb, int s, int lane) {
    const int ln = lane & 15, kg = lane >> 4;
    if (cb == 0) return s < 2 ? A_OFF + 16 * lane : (kg == 0 ? B_OFF + B_REC * ln : ZERO_OFF);
    return s < 2 ? (ln < 8 ? C_OFF + C_REC * (8 * kg + ln) : ZERO_OFF) : (ln < 8 && kg == 0 ? D_OFF + D_REC * ln : ZERO_OFF);
}
XFH_HD constexpr int frag_in_rec(int cb, int s, int q) { return s == 2 ? 16 * q : (cb == 0 ? 1024 * (3 * s + q) : 16 * (3 * s + q)); }
// byte offset of lane's 16 bytes of fragment (cb, s, q)
XFH_HD inline int lane_off(int cb, int s, int q, int lane) { return lane_base(cb, s, lane) + frag_in_rec(cb, s, q); }
XFH_HD inline bool lane_real(int cb, int s, int lane) { return (cb == 0 || (lane & 15) < 8) && (s < 2 || (lane >> 4) == 0); }

// ---- c3 (conv4's input tile: 17 rows x 33 columns x 8 channels, origin (2 Y4 - 1, 2 X4 - 1) of the half-resolution map) as fp16 pairs:
// two planes (high parts, low parts at scale 2^11), a pixel = 16 bytes (its 8 channels), a row = its 17 even columns, then its 16 odd ones:
// the 16 lanes of a B-fragment read step by TWO columns (stride 2) and stay contiguous within a parity.
constexpr int C3H = 17, C3W = 33, C3_NEVEN = 17, C3_ROWB = C3W * 16, C3_PLANE = C3H * C3_ROWB;      // 528, 8976
XFH_HD constexpr int c3_pixel_off(int r, int c) { return r * C3_ROWB + 16 * ((c & 1) ? C3_NEVEN + (c >> 1) : (c >> 1)); }
// B fragment of lane (ln = output column, kg) in K step s of output row orow: the pixel under tap t = min(4 s + kg, 8)
XFH_HD inline int c3_frag_off(int orow, int s, int lane) {
    const int ln = lane & 15, kg = lane >> 4;
    const int t = 4 * s + kg < 8 ? 4 * s + kg : 8;
    return c3_pixel_off(2 * orow + t / 3, 2 * ln + t % 3);
}

// ---- conv3 (block1.2: 8 -> 8, 3x3, stride 1; block1_fused_kernel<7>): 8 couts would leave half of the 16 rows idle, so a column of the product is a PAIR of
// horizontally adjacent output pixels (2 pc, 2 pc + 1) and its K values are the union of their windows: 3 rows x 4 columns x 8 channels = 96 = three K steps of 32
// with nothing padded -- K value 8 kg + j of step s = channel j of the c2 pixel (r + s, 2 pc + kg).  Rows 0 .. 7 = the couts of the left pixel (weights of tap
// (s, kg), zero for kg = 3), rows 8 .. 15 = the couts of the right pixel (tap (s, kg - 1), zero for kg = 0).  D: lane (pair ln, kg) holds couts 4 (kg & 1) + j of the
// pixel 2 pc + (kg >> 1): all 64 lanes carry results.  17 x 17 pairs per tile (the right pixel of the last pair of a row does not exist and is dropped) = 19 blocks
// of 16 pairs in row-major order, 9 MFMAs each.
// Two fragments per step are stored, q0 and q2; q1 = fp16(w) = 2^-11 q0 is derived in the kernel (exact wherever both are normal numbers: |w| >= 2^-14).
//   W3 image: lane-major records of 96 bytes: record 8 dx + cout (dx = 0 .. 2) holds the fragments (s, q0 | q2) at 16 (2 s + j); record 24 = zeros.  Lane (ln, kg)
//   reads record 8 (kg - (ln >> 3)) + (ln & 7), or the zeros where that dx is outside 0 .. 2.
// c2 (conv3's input tile: 19 x 35 x 8, origin (2 Y4 - 2, 2 X4 - 2)) as fp16 pairs: plane of high parts, plane of low parts; a row = its 18 even columns, then its
// 17 odd ones (the 16 lanes of a B fragment step by two columns).
constexpr int W3_REC = 96, W3_ZERO_OFF = 24 * W3_REC, W3_BYTES = 25 * W3_REC, W3_IMAGE_BYTES = 3072;      // 2400 (the HBM image is padded to three 1-KiB pieces)
constexpr int C2H = 19, C2W = 35, C2_NEVEN = 18, C2_ROWB = C2W * 16, C2_PLANE = C2H * C2_ROWB;            // 560, 10640
constexpr int NPAIR = 17, NBLK3 = (C3H * NPAIR + 15) / 16;                                                    // 19 blocks of 16 pairs
XFH_HD constexpr int c2_pixel_off(int r, int c) { return r * C2_ROWB + 16 * ((c & 1) ? C2_NEVEN + (c >> 1) : (c >> 1)); }
// fragment j = 0 (q0), 1 (q2) of K step s
XFH_HD inline int w3_lane_off(int s, int j, int lane) {
    const int dx = (lane >> 4) - ((lane & 15) >> 3);
    return (dx >= 0 && dx <= 2 ? W3_REC * (8 * dx + (lane & 7)) : W3_ZERO_OFF) + 16 * (2 * s + j);
}
// B fragment of lane (pair ln of block blk, kg) in K step s: the c2 pixel (r + s, 2 pc + kg); pair index clamped to the tile, column 35 (pc = 16, kg = 3: it only meets zero
// weights and the dropped pixel) read as column 33
XFH_HD inline int c2_frag_off(int blk, int s, int lane) {
    const int ln = lane & 15, kg = lane >> 4;
    int ep = blk * 16 + ln;
    if (ep > C3H * NPAIR - 1) ep = C3H * NPAIR - 1;
    const int r = ep / NPAIR, pc = ep - r * NPAIR;
    int col = 2 * pc + kg;
    if (col > C2W - 1) col -= 2;
    return c2_pixel_off(r + s, col);
}

// host: folded fp32 weights w_kc[(ci * 9 + tap) * 8 + cout] of conv3 -> the W3 image (W3_IMAGE_BYTES)
template <typename Split>
inline void pack_w3(const float* w_kc, uint16_t* out /* W3_IMAGE_BYTES / 2 */, Split split) {
    for (int i = 0; i < W3_IMAGE_BYTES / 2; ++i) out[i] = 0;
    for (int dx = 0; dx < 3; ++dx)
        for (int co = 0; co < 8; ++co)
            for (int s = 0; s < NSTEP; ++s)            // K step = tap row
                for (int j = 0; j < 8; ++j) {
                    uint16_t q[3];
                    split(w_kc[(j * NTAP + 3 * s + dx) * 8 + co], q);
                    const int rec = W3_REC * (8 * dx + co) / 2;
                    out[rec + 8 * (2 * s) + j] = q[0];
                    out[rec + 8 * (2 * s + 1) + j] = q[2];
                }
}
// host: folded fp32 weights w_kc[(ci * 9 + tap) * 24 + cout] -> the W4_BYTES image above.  split(v, q): the three fp16 fragments of a weight.
template <typename Split>
inline void pack_w4(const float* w_kc, uint16_t* out /* W4_BYTES / 2 */, Split split) {
    for (int i = 0; i < W4_BYTES / 2; ++i) out[i] = 0;
    for (int cb = 0; cb < 2; ++cb)
        for (int s = 0; s < NSTEP; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                if (!lane_real(cb, s, lane)) continue;
                const int co = 16 * cb + (lane & 15), t = 4 * s + (lane >> 4);
                for (int j = 0; j < 8; ++j) {
                    uint16_t q[3];
                    split(w_kc[(j * NTAP + t) * NCO + co], q);
                    for (int k = 0; k < 3; ++k) out[lane_off(cb, s, k, lane) / 2 + j] = q[k];
                }
            }
}

}  // namespace b1fx
}  // namespace xfh
