// Three-way bf16 split of fp32 values (shared by the split-operand MFMA convolutions k_conv_bx.hip / k_conv_bx64.hip).
#pragma once
#ifndef XFH_HOST_EMU      // (tests/emu/ compiles this header for the host)
#include "common.hpp"
#endif

namespace xfh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));


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


// ---- the fp16-pair arithmetic ("fx") ---------------------------------------------------------------------------------------------
// x = xh + 2^-11 xl up to 2^-22 |x|: xh = fp16(x), xl = fp16(2^11 (x - xh)) (round to nearest even; the residual is exact in fp32 and, scaled, of the size of
// x / 2: never a subnormal where xh is not).  With the weights as q0 = fp16(2^11 w), q2 = fp16(2^11 w - q0), q1 = fp16(w) (api.hip: split_weight) a product sum
// takes THREE fp16 MFMAs per K = 16 instead of the six of the bf16 three-way split,
//     2^11 w x  =  q2 xh + q1 xl + q0 xh   (+ the dropped term 2^-11 (2^11 w - q0 - q2) ... <= 2^-21 |w x| in all),
// every partial product exact in the fp32 accumulator, ONE accumulator at scale 2^11 (the epilogue multiplies by 2^-11, exactly).  Error of a K = 576 product
// sum against fp64: 2.7e-7 of max |y| in the numpy restatement (the six-MFMA bf16 form: 5.5e-7; an fp32 fma chain: 7.3e-7) -- fewer roundings of the accumulator.
// Range: |x| < 65504 (detected: FxRange), |w| < 31 (checked at xfh_create: the layer otherwise stays on the bf16 form).
__device__ inline void split2_f16(float a, float b, unsigned& h, unsigned& l) {
    const f32x2 v = {a, b};
    const f16x2 hh = __builtin_convertvector(v, f16x2);                       // v_cvt_pk_f16_f32 (round to nearest even)
    const f32x2 r = (v - __builtin_convertvector(hh, f32x2)) * 2048.f;
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
// The same bits from scalar operations (conv_rs64_body.hpp: a lone wave per SIMD pays three issue slots for a packed fp32 instruction beside its MFMA stream): the residual
// 2^11 (x - xh) = fma(xh, -2^11, 2^11 x) is exact in fp32, as the difference above is, so the low parts agree bit for bit; hipcc selects v_fma_mix_f32 (the fp16 high part
// as an operand: no conversion back)
__device__ inline void split2_f16_scalar(float a, float b, unsigned& h, unsigned& l) {
    const f32x2 v = {a, b};
    const f16x2 hh = __builtin_convertvector(v, f16x2);
    const f32x2 r = {__builtin_fmaf((float)hh[0], -2048.f, a * 2048.f), __builtin_fmaf((float)hh[1], -2048.f, b * 2048.f)};
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}
constexpr float FX_SCALE_INV = 1.f / 2048.f;
// Range guard of the fp16 pair: a kernel keeps the largest |x| it converted (one v_max3 per value pair) and, when it ends, reports |x| >= 65504 -- a value the
// fp16 high part cannot hold -- by setting bit 0 of the caller's status word (xfh_set_status_buffer; the host re-runs the batch in the bf16 arithmetic).
constexpr float FX_MAX_INPUT = 65504.f;
__device__ inline void fx_track(float& amax, float a, float b) { amax = fmaxf(amax, fmaxf(fabsf(a), fabsf(b))); }
// The same on the converted HIGH PARTS, for kernels where the float form costs too much (k_heads.hip: tracking the fp32 values kept them alive next to their
// fragments: + 44 registers, + 300 instructions): x left the fp16 range exactly when its high part became inf (0x7c00; NaN above), and as unsigned 16-bit
// integers the magnitudes of fp16 numbers order like their values -- one v_pk_max_u16 per converted pair (sign bits masked off where the values are signed).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ inline void fx_track_h(unsigned& amax, unsigned h, bool is_signed) {
    if (is_signed) h &= 0x7fff7fffu;
    amax = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2, amax), __builtin_bit_cast(u16x2, h)));
}
__device__ inline void fx_report_h(unsigned amax, int* status) {
    if (status && ((amax & 0xffffu) >= 0x7c00u || (amax >> 16) >= 0x7c00u)) atomicOr(status, 1);
}
__device__ inline void fx_report(float amax, int* status) {
    if (status && amax >= FX_MAX_INPUT) atomicOr(status, 1);
}

}  // namespace xfh
