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

}  // namespace xfh
