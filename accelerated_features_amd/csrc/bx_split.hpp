// The fp16-pair arithmetic: device side (shared by every kernel that runs fp32 product sums on the fp16 matrix cores).
#pragma once
#ifndef XFH_HOST_EMU      // (tests/emu/ compiles this header for the host)
#include "common.hpp"
#endif

namespace xfh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));


// ---- the fp16-pair arithmetic ("fx") ---------------------------------------------------------------------------------------------
// x = xh + 2^-11 xl up to 2^-22 |x|: xh = fp16(x), xl = fp16(2^11 (x - xh)) (round to nearest even; the residual is exact in fp32 and, scaled, of the size of
// x / 2: never a subnormal where xh is not).  With the weights as q0 = fp16(2^11 w), q2 = fp16(2^11 w - q0), q1 = fp16(w) (api.hip: split_weight) a product sum
// takes THREE fp16 MFMAs per K = 16,
//     2^11 w x  =  q2 xh + q1 xl + q0 xh   (+ the dropped term 2^-11 (2^11 w - q0 - q2) ... <= 2^-21 |w x| in all),
// every partial product exact in the fp32 accumulator, ONE accumulator at scale 2^11 (the epilogue multiplies by 2^-11, exactly).  Error of a K = 576 product
// sum against fp64: 2.7e-7 of max |y| in the numpy restatement (an fp32 fma chain: 7.3e-7) -- fewer roundings of the accumulator.
// Range: |x| < 65504 (detected: FxRange), |w| < 31 (checked at xfh_create: the layer otherwise runs on its f32-MFMA kernel).
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
// fp16 high part cannot hold -- by setting bit 0 of the caller's status word (xfh_set_status_buffer; the host re-runs the batch on the f32-MFMA / vector-ALU kernels).
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
