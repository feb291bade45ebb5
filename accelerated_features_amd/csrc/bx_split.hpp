// Three-way bf16 split of fp32 values (shared by the split-operand MFMA convolutions k_conv_bx.hip / k_conv_bx64.hip).
#pragma once
#include "common.hpp"

namespace xfh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));


__device__ inline unsigned pk_bf16_rne(float a, float b) {      // v_cvt_pk_bf16_f32: a -> low half
    const f32x2 v = {a, b};
    const bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(unsigned, r);
}
// (a, b) -> packed bf16 pairs h, m, l with a = ah + am + al (+ 2^-27 |a|)
__device__ inline void split3(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = pk_bf16_rne(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = pk_bf16_rne(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
    l = pk_bf16_rne(sa, sb);
}

// (Measured in round 3 and dropped: the residual r = x - h as ONE v_dot2c_f32_bf16 on the packed word with the constant pair (-1, 0) / (0, -1) -- 7 ops per
// pair instead of 11, exact, all 75 GPU tests green as the split of every split-bf16 kernel -- but no kernel got faster and k_conv_bx64s2, whose split sits inside its
// MFMA rows, got 10 % slower: the dot instruction does not issue at the rate of an and / sub.  Also: as a literal, hipcc encodes the pair 0x0000bf80 as the inline
// constant -1.0, which the instruction reads as the fp32 pattern 0xbf800000 = the pair (0, -1).)
// The same by truncation: h, m = the leading 8 + 8 significant bits, l = the remaining 8 -- a + b + c is EXACT (fp32 has 24), every op is a
// plain and / sub / v_perm_b32 (which packs the two high halves).  |m| < 2^-7 |a|,
// |l| < 2^-15 |a|: with RNE-split weights (|wm| <= 2^-9, |wl| <= 2^-18) the three dropped cross terms stay below 2^-23 of the product, zero-mean.
__device__ inline void split3_trunc(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
    const float ra = a - __uint_as_float(__float_as_uint(a) & 0xffff0000u), rb = b - __uint_as_float(__float_as_uint(b) & 0xffff0000u);
    m = __builtin_amdgcn_perm(__float_as_uint(rb), __float_as_uint(ra), 0x07060302u);
    const float sa = ra - __uint_as_float(__float_as_uint(ra) & 0xffff0000u), sb = rb - __uint_as_float(__float_as_uint(rb) & 0xffff0000u);
    l = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
}

}  // namespace xfh
