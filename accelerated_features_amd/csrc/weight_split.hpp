// Host side of the fp16-pair weight formats (api.hip packs every MFMA operand with these; tests/test_block1_fx_model.py compiles them with g++).
#pragma once
#include <cstdint>
#include <cstring>

namespace xfh {

// fp32 -> fp16, round to nearest even (subnormals kept, overflow -> inf) and back: the host side of the two-term fp16 weights
inline uint16_t f16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u) return (uint16_t)(sign | (u > 0x7f800000u ? 0x7e00u : 0x7c00u));      // NaN / inf
    if (u >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                       // >= 65520: rounds to inf
    if (u < 0x38800000u) {                                                                        // < 2^-14: subnormal result, spacing 2^-24
        if (u < 0x33000000u) return (uint16_t)sign;                                               // < 2^-25: rounds to zero (2^-25 itself ties to even = 0)
        const int e = (int)(u >> 23);                                                             // biased fp32 exponent, 102 .. 112
        const uint32_t m = (u & 0x7fffffu) | 0x800000u;                                           // 24-bit significand
        const int sh = 126 - e;                                                                   // value = m * 2^(e - 150); result units of 2^-24: m >> (126 - e)
        const uint32_t q = m >> sh, rem = m & ((1u << sh) - 1u), halfway = 1u << (sh - 1);
        return (uint16_t)(sign | (q + ((rem > halfway || (rem == halfway && (q & 1u))) ? 1u : 0u)));
    }
    const uint32_t v = u - 0x38000000u;                                                           // rebias 127 -> 15
    return (uint16_t)(sign | ((v + 0xfffu + ((v >> 13) & 1u)) >> 13));
}
inline float f16_float(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
    uint32_t u;
    if (e == 0) {
        const float f = (float)m * 5.9604644775390625e-8f;                                        // m * 2^-24, exact
        memcpy(&u, &f, 4);
        u |= sign;
    } else if (e == 31) u = sign | 0x7f800000u | (m << 13);
    else u = sign | ((e + 112u) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}
// The three fp16 fragments of one fp32 weight in the fp16-pair arithmetic (bx_split.hpp), all at scale 2^11: q0 = fp16(2^11 w), q2 = fp16(2^11 w - q0) -- together 22 bits
// of 2^11 w, multiplied with the activation's high part -- and q1 = fp16(w), multiplied with the activation's 2^11-scaled low part.
inline void split_weight(float v, uint16_t (&q)[3]) {
    const float s = v * 2048.f;                    // exact
    q[0] = f16_rne(s);
    q[1] = f16_rne(v);
    q[2] = f16_rne(s - f16_float(q[0]));            // exact difference
}
constexpr float kFxMaxWeight = 31.f;                   // |w| * 2^11 must stay below the fp16 maximum (65504)

// A linear layer y = x W for linear_fx_kernel (k_linear_mfma.hip: the fine_matcher's layers in the fp16-pair arithmetic): w_kn [K][n_pad] fp32 (BatchNorm folded, zero padded)
// -> [column block of 64][K step of 16][fragment 3][column half-block 2][lane = half * 32 + column][8]: lane (column n = 64 nb + 32 cb + (lane & 31), half) holds k = 16 s + 8 half + i.
// Returns the 16-bit words written (3 K n_pad).
inline size_t pack_linear_fx(const float* w_kn, int K, int n_pad, uint16_t* dst) {
    const int nbs = n_pad / 64, ns = K / 16;
    for (int nb = 0; nb < nbs; ++nb)
        for (int st = 0; st < ns; ++st)
            for (int cb = 0; cb < 2; ++cb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 8; ++i) {
                        const int k = 16 * st + 8 * (lane >> 5) + i, n = 64 * nb + 32 * cb + (lane & 31);
                        uint16_t q[3];
                        split_weight(w_kn[(size_t)k * n_pad + n], q);
                        for (int sp = 0; sp < 3; ++sp) dst[(((((size_t)nb * ns + st) * 3 + sp) * 2 + cb) * 64 + lane) * 8 + i] = q[sp];
                    }
    return (size_t)3 * K * n_pad;
}

// One layer of a head (head_bx_body.hpp) in operand order: [K step t][cout block][fragment][lane = half * 32 + cout][8], cout blocks of 32 (zeros above cout).
// K order: the first layer of a head takes its 64 input channels in natural order (16 t + 8 half + i); a chained layer takes the previous layer's D registers, i.e.
// feature 32 (t >> 1) + 16 (t & 1) + 8 (i >> 2) + 4 half + (i & 3).  w: (cout, 64) fp32 (BatchNorm folded).  Returns the 16-bit words written.
inline size_t pack_head_layer(const float* w, int cout, bool first, uint16_t* dst) {
    const int mbo = (cout + 31) / 32;
    for (int t = 0; t < 4; ++t)
        for (int mb = 0; mb < mbo; ++mb)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i) {
                    const int o = mb * 32 + (lane & 31), hf = lane >> 5;
                    const int ch = first ? 16 * t + 8 * hf + i : 32 * (t >> 1) + 16 * (t & 1) + 8 * (i >> 2) + 4 * hf + (i & 3);
                    uint16_t q[3];
                    split_weight(o < cout ? w[(size_t)o * 64 + ch] : 0.f, q);
                    for (int sp = 0; sp < 3; ++sp) dst[((((size_t)t * mbo + mb) * 3 + sp) * 64 + lane) * 8 + i] = q[sp];
                }
    return (size_t)4 * mbo * 3 * 64 * 8;
}

// 3x3 stride-2 convolution weights for conv_bx64s2x_kernel: [cout half][cin/16][tap 9][cout block 2][fragment 3][lane = half * 32 + cout][8], channel = 16 chunk + 8 half + i.
// w: (cout, cin, 3, 3) fp32 (BatchNorm folded), cin % 16 == 0, cout % 64 == 0.  Returns the 16-bit words written.
inline size_t pack_bx64(const float* w, int cin, int cout, uint16_t* dst) {
    const int nch = cin / 16, nhf = cout / 64;
    for (int hf = 0; hf < nhf; ++hf)
        for (int ch = 0; ch < nch; ++ch)
            for (int tap = 0; tap < 9; ++tap)
                for (int cb = 0; cb < 2; ++cb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int i = 0; i < 8; ++i) {
                            const int o = hf * 64 + cb * 32 + (lane & 31), ci = ch * 16 + 8 * (lane >> 5) + i;
                            uint16_t q[3];
                            split_weight(w[((size_t)o * cin + ci) * 9 + tap], q);
                            for (int sp = 0; sp < 3; ++sp) dst[((((((size_t)hf * nch + ch) * 9 + tap) * 2 + cb) * 3 + sp) * 64 + lane) * 8 + i] = q[sp];
                        }
    return (size_t)nhf * nch * 9 * 2 * 3 * 64 * 8;
}
// 3x3 convolution weights (64 -> 64) for conv_rs64_kernel (conv_rs64_body.hpp: weights resident in registers, K split over the four waves of a workgroup):
// [wave = 16-channel chunk][tap 9][cout block 2][fragment 3][lane = half * 32 + cout][8], channel = 16 wave + 8 half + i;
// w: (64, 64, 3, 3) fp32 (BatchNorm folded).  Returns the 16-bit words written.
constexpr size_t kRs64Halfs = (size_t)4 * 9 * 2 * 3 * 64 * 8;      // 216 KiB
inline size_t pack_rs64(const float* w, uint16_t* dst) {
    for (int wv = 0; wv < 4; ++wv)
        for (int tap = 0; tap < 9; ++tap)
            for (int cb = 0; cb < 2; ++cb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 8; ++i) {
                        const int o = cb * 32 + (lane & 31), ci = wv * 16 + 8 * (lane >> 5) + i;
                        uint16_t q[3];
                        split_weight(w[((size_t)o * 64 + ci) * 9 + tap], q);
                        for (int sp = 0; sp < 3; ++sp) dst[(((((size_t)wv * 9 + tap) * 2 + cb) * 3 + sp) * 64 + lane) * 8 + i] = q[sp];
                    }
    return kRs64Halfs;
}
// the 128 -> 128 form of conv_rs64_kernel: [cout quarter 4][wave = 32-channel group 4][tap 9][chunk 2][fragment 3][lane = half * 32 + cout][8],
// channel = 32 wave + 16 chunk + 8 half + i, cout = 32 quarter + (lane & 31).  w: (128, 128, 3, 3) fp32.  Returns the 16-bit words written (4 x kRs64Halfs).
inline size_t pack_rs128(const float* w, uint16_t* dst) {
    for (int cq = 0; cq < 4; ++cq)
        for (int wv = 0; wv < 4; ++wv)
            for (int tap = 0; tap < 9; ++tap)
                for (int ck = 0; ck < 2; ++ck)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int i = 0; i < 8; ++i) {
                            const int o = cq * 32 + (lane & 31), ci = wv * 32 + ck * 16 + 8 * (lane >> 5) + i;
                            uint16_t q[3];
                            split_weight(w[((size_t)o * 128 + ci) * 9 + tap], q);
                            for (int sp = 0; sp < 3; ++sp) dst[(((((((size_t)cq * 4 + wv) * 9 + tap) * 2 + ck) * 3 + sp) * 64) + lane) * 8 + i] = q[sp];
                        }
    return 4 * kRs64Halfs;
}
// the 1x1 (64 -> 64) fused behind the 3x3 in conv_rs64_kernel: A operands of v_mfma_f32_16x16x32_f16, [wave = couts 16 wave .. + 15][K step 2][fragment 3][lane = (K group l >> 4, cout l & 15)][8],
// channel = 32 step + 8 (l >> 4) + i (the natural order: the kernel lays the 3x3's outputs out that way).  w: (64, 64) fp32.
inline size_t pack_rs64_1x1(const float* w, uint16_t* dst) {
    for (int wv = 0; wv < 4; ++wv)
        for (int s = 0; s < 2; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i) {
                    uint16_t q[3];
                    split_weight(w[(size_t)(16 * wv + (lane & 15)) * 64 + 32 * s + 8 * (lane >> 4) + i], q);
                    for (int sp = 0; sp < 3; ++sp) dst[((((size_t)wv * 2 + s) * 3 + sp) * 64 + lane) * 8 + i] = q[sp];
                }
    return (size_t)4 * 2 * 3 * 64 * 8;
}
}  // namespace xfh
