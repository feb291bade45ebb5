// The split-operand heads (head_bx_kernel, k_heads.hip): the kernel body in a header of its own so that tests/emu/ can compile the SAME source for the host
// (XFH_HOST_EMU) and run it against a float64 reference without a GPU.
#pragma once
#ifndef XFH_HOST_EMU
#include "kernels.hpp"
#include <type_traits>
#define XFH_DYN_LDS_BYTES(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#define XFH_LDS_VOLATILE(T) __attribute__((address_space(3))) volatile T
#ifndef XFH_PIN
#define XFH_PIN(x) asm volatile("" : "+v"(x))
#endif
#ifndef XFH_GPTR_DEFINED
#define XFH_GPTR_DEFINED
typedef __attribute__((address_space(1))) const void* xfh_gptr_t;
typedef __attribute__((address_space(3))) void* xfh_lptr_t;
#endif
#endif
#include "bx_split.hpp"

namespace xfh {

#ifndef XFH_HD_CELLS
#define XFH_HD_CELLS
constexpr int HD_CELLS = 256;    // cells per tile
#endif

// ------------------------------------------------------------------------------------------------------------------------------
// The same heads on the bf16 matrix cores with three-way split operands (k_conv_bx.hip has the arithmetic: six
// v_mfma_f32_32x32x16_bf16 per K = 16 carry an fp32 product sum at 3/8 of the f32-MFMA pipe time).
//
// A head is a chain of K = 64 layers on a wave's 32 cells, so nothing is staged: the split weights of every layer sit in LDS in operand
// order (111 KiB for the key-point head), the activations stay in registers -- a layer's D registers (lane = cell, registers = features
// (r & 3) + 8 (r >> 2) + 4 half) become the next layer's B fragments once ReLU'd and split (K step t of lane half h takes the register
// quads 8 (t & 1) and 8 (t & 1) + 4 of block t >> 1: features 32 (t >> 1) + 16 (t & 1) + 8 q + 4 h + i; the weights are packed in that K
// order) -- and the first layer's eight consecutive channels per K step are 32 contiguous bytes of the source (one pixel row of the 8x8
// cell for the unfold, one slice of the channels-last feature row), loaded straight into registers one tile ahead.  No barrier after the
// weights have landed: the eight waves of a workgroup drift apart and fill each other's split / softmax phases.
// ------------------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct HeadBxArgs {
    const float* src;        // KP: raw gray (B,H,W) ; REL: feats (B*hc*wc, 64)
    const float* coef;       // KP: per-image instance-norm {alpha, beta}
    const uint4* wq;         // all layers: [layer][K step 4][cout block][split 3][64 lanes] 8 bf16   (api.hip: pack_head_bx)
    const float* bias;       // all layers' biases, padded to the cout blocks: KP 64,64,64,96 ; REL 64,64
    const float* w_last;     // REL: the 64 weights of the final 64 -> 1 layer
    float b_last;            //      and its bias
    float* out;              // KP: heat (B,H,W) ; REL: reliability (B*hc*wc)
    float* logits;           // KP only, optional: (B*hc*wc, 65)
    float* inv;              // REL only, optional: 1 / max(||feats[cell,:]||, 1e-12)
    int H, W, hc, wc, ncell, ntiles;
    long long* trace;        // debug: s_memtime stamps of wave 0's second tile, 16 per workgroup
    int cold;                // debug (xfh_debug_cold_start)
    int* status;             // fx: range guard (bx_split.hpp), may be NULL
    const float* w_dust;     // KP, fx forms: the 64 fp32 weights of the dustbin logit (row 64 of keypoint_head.3) -- computed as a dot product on the vector ALUs,
    float b_dust;            //     and its bias: the last layer then has two cout blocks instead of three (12 of 36 MFMAs and 16 accumulator registers less)
};

__device__ inline unsigned hb_pk_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// eight fp32 values -> the three bf16 fragments h, m, l (x = h + m + l up to 2^-27 |x|; round to nearest even, exact residuals)
__device__ inline void hb_split8(const float (&y)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    uint4 uh, um, ul;
    unsigned* ph = &uh.x; unsigned* pm = &um.x; unsigned* pl = &ul.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = y[2 * i], b = y[2 * i + 1];
        const unsigned hh = hb_pk_bf16(a, b);
        const float ra = a - __uint_as_float(hh << 16), rb = b - __uint_as_float(hh & 0xffff0000u);
        const unsigned mm = hb_pk_bf16(ra, rb);
        const float sa = ra - __uint_as_float(mm << 16), sb = rb - __uint_as_float(mm & 0xffff0000u);
        ph[i] = hh; pm[i] = mm; pl[i] = hb_pk_bf16(sa, sb);
    }
    h = __builtin_bit_cast(bf16x8, uh); m = __builtin_bit_cast(bf16x8, um); l = __builtin_bit_cast(bf16x8, ul);
}

// one K = 64 layer: out[mb] = bias + W x, x given per K step by `xs(t, y[8])`; weights of the layer at wl (LDS, operand order)
// FX: the fp16-pair arithmetic (bx_split.hpp): two input fragments, three MFMAs per K step and accumulator, `out` at scale 2^11 WITHOUT the bias (the consumer
// applies fma(out, 2^-11, bias)); amax collects the largest |x| converted (range guard)
// FXM: 0 = bf16 three-way split; 1 = fp16 pair; 2 = fp16 pair with TWO weight fragments in LDS (q0, q2; q1 = fp16(w) = 2^-11 q0 is derived with four v_pk_mul_f16:
// a third less LDS and LDS read traffic for + 8 % vector instructions); 3 = 1 + the B fragments take a round trip through a wave-private LDS slot (`lbuf`, 4 KiB per
// wave) before the matrix core sees them: EVERY operand register is then written by the LDS return path, as in every convolution kernel -- none of which ever tripped on
// instruction-cache refills -- and none by the vector ALU.  An experiment for the cold-start scan, paid with 4 LDS operations per K step.
template <int MBO, int FXM, bool RELU_IN, typename XS>      // RELU_IN: xs delivers non-negative values (a chained layer): cheaper range tracking
__device__ inline void head_bx_layer(const unsigned char* wl, const float* bias_lds, XS xs, f32x16 (&out)[MBO], int lane, int half, unsigned& amax, unsigned char* lbuf) {
    constexpr bool FX = FXM > 0;
    constexpr int NW = FXM == 2 ? 2 : 3;      // weight fragments per (K step, cout block) in LDS
    using frag_t = std::conditional_t<FX, f16x8, bf16x8>;
    constexpr int NXS = FX ? 2 : 3;
#pragma unroll
    for (int mb = 0; mb < MBO; ++mb)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 t = FX ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(bias_lds + mb * 32 + 8 * g4 + 4 * half);
            out[mb][4 * g4] = t.x; out[mb][4 * g4 + 1] = t.y; out[mb][4 * g4 + 2] = t.z; out[mb][4 * g4 + 3] = t.w;
        }
    // (compiler fence: the weights never change, so hipcc hoists every fragment read of every layer out of the persistent tile loop --
    // 432 registers' worth, straight into scratch memory)
    asm volatile("" ::: "memory");
    frag_t w[2][MBO][3];
    auto ldw = [&](int t, frag_t (&o)[MBO][3]) {
#pragma unroll
        for (int mb = 0; mb < MBO; ++mb)
#pragma unroll
            for (int q = 0; q < NW; ++q) o[mb][NW == 3 ? q : 2 * q] = *reinterpret_cast<const frag_t*>(wl + (((t * MBO + mb) * NW + q) * 64 + lane) * 16);
    };
    auto derive = [&](frag_t (&o)[MBO][3]) {      // (two fragments stored: q1 from q0, exact wherever both are normal numbers)
        if constexpr (NW == 2) {
#pragma unroll
            for (int mb = 0; mb < MBO; ++mb) o[mb][1] = o[mb][0] * (_Float16)0.00048828125f;
        }
    };
    ldw(0, w[0]);
    derive(w[0]);
    // The B fragments are VALU results, and a VALU write that follows an MFMA by a few cycles can land in that MFMA's A/B registers
    // before the matrix core has read all 64 lanes of them (seen here: the cells of lanes 16-31 of a wave wrong in a few launches out
    // of many; hipcc's hazard recogniser only covers SrcC).  So the fragments of step t+1 are built into a second register set while
    // step t's are still alive: the empty asm below is a use of step t's set AFTER the split, which keeps the allocator from handing
    // its registers to the new values; a set is rewritten one full step (12-18 MFMAs) after its last read.
    frag_t xf[2][NXS];
    auto split8 = [&](const float (&y)[8], frag_t (&o)[NXS], int slot) {
        if constexpr (FX) {
            uint4 uh, ul;
            unsigned* ph = &uh.x; unsigned* pl = &ul.x;
#pragma unroll
            for (int i = 0; i < 4; ++i) { split2_f16(y[2 * i], y[2 * i + 1], ph[i], pl[i]); fx_track_h(amax, ph[i], !RELU_IN); }
            o[0] = __builtin_bit_cast(frag_t, uh); o[1] = __builtin_bit_cast(frag_t, ul);
            if constexpr (FXM == 3) {      // through the wave's LDS slot and back (volatile: the store and the load are both emitted)
                typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
                XFH_LDS_VOLATILE(u32x4v)* p = (XFH_LDS_VOLATILE(u32x4v)*)(lbuf + slot * 2048 + lane * 16);      // (an LDS pointer by type: ds_write / ds_read, not flat accesses)
                p[0] = __builtin_bit_cast(u32x4v, o[0]); p[64] = __builtin_bit_cast(u32x4v, o[1]);
                const u32x4v rh = p[0], rl = p[64];
                o[0] = __builtin_bit_cast(frag_t, rh); o[1] = __builtin_bit_cast(frag_t, rl);
            }
        } else {
            hb_split8(y, o[0], o[1], o[NXS - 1]);
        }
    };
    {
        float y[8];
        xs(0, y);
        split8(y, xf[0], 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t + 1 < 4) ldw(t + 1, w[(t + 1) & 1]);
        asm volatile("" ::: "memory");
        const frag_t x0 = xf[t & 1][0], x1 = xf[t & 1][1], x2 = xf[t & 1][NXS - 1];
        __builtin_amdgcn_sched_barrier(0);      // the MFMAs of a step stay together: left free, hipcc floats the NEXT steps' splits in between them
        // products (weight split, input split), small terms first: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h); fp16 pair: (q2, xh) (q1, xl) (q0, xh)
#define HB_MM(WQ, X) { _Pragma("unroll") for (int mb = 0; mb < MBO; ++mb) { \
            if constexpr (FX) out[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[t & 1][mb][WQ], X, out[mb], 0, 0, 0); \
            else out[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[t & 1][mb][WQ], X, out[mb], 0, 0, 0); } }
        if constexpr (FX) { HB_MM(2, x0) HB_MM(1, x1) HB_MM(0, x0) }
        else { HB_MM(2, x0) HB_MM(0, x2) HB_MM(1, x1) HB_MM(1, x0) HB_MM(0, x1) HB_MM(0, x0) }
#undef HB_MM
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < 4) {
            float y[8];
            xs(t + 1, y);
            split8(y, xf[(t + 1) & 1], (t + 1) & 1);
            derive(w[(t + 1) & 1]);
            // the new fragments pass THROUGH the asm that uses the old ones (and this step's weights): it cannot move above the split,
            // so the old registers stay occupied while the split's results and temporaries are written
#ifndef XFH_HOST_EMU
            asm volatile("" : "+v"(xf[(t + 1) & 1][0]), "+v"(xf[(t + 1) & 1][1]), "+v"(xf[(t + 1) & 1][NXS - 1])
                            : "v"(xf[t & 1][0]), "v"(xf[t & 1][1]), "v"(xf[t & 1][NXS - 1]), "v"(w[t & 1][0][0]), "v"(w[t & 1][0][1]), "v"(w[t & 1][0][2]),
                              "v"(w[t & 1][MBO - 1][0]), "v"(w[t & 1][MBO - 1][1]), "v"(w[t & 1][MBO - 1][2]));
            if (MBO == 3) asm volatile("" : "+v"(xf[(t + 1) & 1][0]) : "v"(w[t & 1][1][0]), "v"(w[t & 1][1][1]), "v"(w[t & 1][1][2]));
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // Nothing keeps the last step's operand registers from whatever VALU code follows the layer (the next tile's address arithmetic, the next layer's
    // split): idle cycles.  One MFMA's worth (32) was enough while every wave on the SIMD ran this kernel; with a second batch in flight on another
    // stream (FrameStream) a wave of ANOTHER kernel -- 64-cycle f32 MFMAs, its own register traffic -- shares the SIMD, the operand fetch of lanes
    // 16-31 comes later, and one 16-cell block of the heat map in ~6000 concurrent steps was wrong (tools/lanes_backbone_soak.py).  128 cycles.
    __builtin_amdgcn_sched_barrier(0);
#ifndef XFH_HOST_EMU
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                 "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7");
#endif
    __builtin_amdgcn_sched_barrier(0);
}

// NOT on the default path since round 4 (option heads_f32 = 0 selects it): with a cold instruction cache -- other kernels evicting its code between launches, or
// xfh_debug_cold_start -- the FIRST tile of a workgroup comes out with the cells of lanes 16..31 of one wave wrong once in 10^3 .. 10^5 launches, depending on
// where the 64-byte instruction lines fall in the MFMA groups (tools/head_soak.py scans 16 code positions: three fail) and on the chip; not understood at the
// instruction level (profiles/r04_head_hazard/).  SHIFT moves the body by 4 x SHIFT bytes for that scan.
// FX: the fp16-pair arithmetic (bx_split.hpp; weights a.wq = NetWeights::head_fx): three MFMAs per K step and cout block instead of six; a layer's accumulators
// live at scale 2^11 without the bias, the consumer applies fma(acc, 2^-11, bias) (the chain in front of its ReLU, the epilogues on the logits)
template <bool KP, int FXM>
__device__ __forceinline__ void head_bx_body(const HeadBxArgs& a) {
    constexpr bool FX = FXM > 0;
    constexpr int NW = FXM == 2 ? 2 : 3;
    constexpr int NB = KP ? 288 : 128;                                 // bias floats
    constexpr bool DUST = KP && FX;                                   // the dustbin logit as a dot product (HeadBxArgs::w_dust)
    constexpr int W_BYTES = (KP ? (DUST ? 8 : 9) : 4) * 4 * NW * 1024;                     // cout blocks x K steps x fragments x 1 KiB
    constexpr int L_BYTES = 2 * 4 * NW * 1024;                        // a 64 -> 64 layer
    XFH_DYN_LDS_BYTES(smem_h);
    float* bias_lds = reinterpret_cast<float*>(smem_h + W_BYTES);
    float* dust_lds = bias_lds + NB;                                   // DUST: the 64 dustbin weights
    unsigned char* lbuf = smem_h + W_BYTES + (NB + 64) * 4 + (threadIdx.x >> 6) * 4096;      // FXM 3: this wave's slot pair
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = a.hc * a.wc;
    for (int j = wave; j < W_BYTES / 1024; j += 8)
        __builtin_amdgcn_global_load_lds((xfh_gptr_t)(reinterpret_cast<const unsigned char*>(a.wq) + j * 1024 + lane * 16), (xfh_lptr_t)(smem_h + j * 1024), 16, 0, 0);
    if (tid < NB) bias_lds[tid] = a.bias[tid];
    if (DUST && tid >= 448) dust_lds[tid - 448] = a.w_dust[tid - 448];

    // first-layer input of this lane's cell: K step t, lane half h = channels 16 t + 8 h .. + 7 = 32 contiguous bytes
    float xin[4][8];
    float nalpha = 1.f, nbeta = 0.f;                                   // (of the tile xin belongs to)
    auto issue_x = [&](int tile) __attribute__((always_inline)) {
        const int g = min(tile * HD_CELLS + wave * 32 + l31, a.ncell - 1);      // cells past the end: copies of the last one, never stored
        const float* p;
        size_t step;
        if (KP) {
            const int b = g / hw, rem = g - b * hw;
            const int ci = rem / a.wc, cj = rem - ci * a.wc;
            p = a.src + (size_t)b * a.H * a.W + (size_t)(8 * ci + half) * a.W + 8 * cj;       // pixel row dy = 2 t + h of the cell
            step = 2 * (size_t)a.W;
            nalpha = a.coef[2 * b]; nbeta = a.coef[2 * b + 1];
        } else {
            p = a.src + (size_t)g * 64 + 8 * half;
            step = 16;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 u0 = *reinterpret_cast<const float4*>(p + t * step), u1 = *reinterpret_cast<const float4*>(p + t * step + 4);
            xin[t][0] = u0.x; xin[t][1] = u0.y; xin[t][2] = u0.z; xin[t][3] = u0.w;
            xin[t][4] = u1.x; xin[t][5] = u1.y; xin[t][6] = u1.z; xin[t][7] = u1.w;
        }
    };

    int tile = blockIdx.x;
    if (tile < a.ntiles) issue_x(tile);
    lds_dma_barrier();                                                // the weights (and biases) have landed; no barrier from here on
    unsigned amax = 0;                                                // fx: the largest fp16 high parts converted, as a pair of 16-bit magnitudes (range guard: bx_split.hpp)
    // fx: the lane's view of the bias table (+ 4 half), as ONE opaque 32-bit LDS offset plus small constants: left to itself hipcc gives every read of the table an
    // address register of its own (the table sits beyond the 64-KiB reach of an offset field), hoisted out of the tile loop and spilled; and the offset is pinned as an
    // INTEGER so that the pointer keeps its LDS type (a pinned pointer becomes generic: flat loads)
    int bias_off = W_BYTES + 16 * half;
    XFH_PIN(bias_off);
    const float* bias_v = reinterpret_cast<const float*>(smem_h + bias_off);                                  // an address register of its own (the table sits beyond the 64-KiB reach of an offset field), hoisted out of the tile loop and spilled
    long long* tr = FXM < 2 && a.trace && tid == 0 ? a.trace + (size_t)blockIdx.x * 16 : nullptr;      // (the experimental forms have no registers to spare for the stamps)
    int tix = 0;
#define HB_STAMP(k) { if (tr && tix == 1) tr[k] = __builtin_amdgcn_s_memtime(); }
    for (; tile < a.ntiles; tile += gridDim.x, ++tix) {
        const int gcell = tile * HD_CELLS + wave * 32 + l31;          // this lane's cell
        HB_STAMP(0)
        f32x16 accA[2], accB[2];
        float nrm2 = 0.f;
        {
            const float al = nalpha, be = nbeta;
            head_bx_layer<2, FXM, false>(smem_h, bias_lds, [&](int t, float (&y)[8]) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    y[i] = KP ? fmaf(xin[t][i], al, be) : xin[t][i];
                    if (!KP) nrm2 = fmaf(y[i], y[i], nrm2);
                }
            }, accA, lane, half, amax, lbuf);
        }
        if (!KP && a.inv) {
            nrm2 += xhalf(nrm2);                                      // the other 32 channels sit in the other half-wave
            if (half == 0 && gcell < a.ncell) a.inv[gcell] = 1.f / fmaxf(sqrtf(nrm2), 1e-12f);
        }
        HB_STAMP(1)
        constexpr bool LATE_X = false;                // (a form without 32 registers to spare during the chained layers would prefetch under the softmax only: none needs it since the dustbin logit left the matrix cores)
        if (!LATE_X && tile + (int)gridDim.x < a.ntiles) issue_x(tile + gridDim.x);      // the next tile's input flies during the chained layers
        // chained layers: K step t = register quads 8 (t & 1), 8 (t & 1) + 4 of block t >> 1, ReLU'd
        // (fx: `in` is at scale 2^11 and without its bias: both applied here -- the biases of the lane's eight features are two float4 of the LDS table)
        auto chain = [](const f32x16 (&in)[2], const float* bias_in) {      // bias_in: this lane's view of the table (+ 4 half)
            return [&in, bias_in](int t, float (&y)[8]) {
                if constexpr (FX) {
                    const float4 b0 = *reinterpret_cast<const float4*>(bias_in + (t >> 1) * 32 + 16 * (t & 1));
                    const float4 b1 = *reinterpret_cast<const float4*>(bias_in + (t >> 1) * 32 + 16 * (t & 1) + 8);
                    const float bq[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) y[i] = fmaxf(fmaf(in[t >> 1][8 * (t & 1) + i], FX_SCALE_INV, bq[i]), 0.f);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) y[i] = fmaxf(in[t >> 1][8 * (t & 1) + i], 0.f);
                }
            };
        };
        if (KP) {
            head_bx_layer<2, FXM, true>(smem_h + L_BYTES, bias_lds + 64, chain(accA, bias_v), accB, lane, half, amax, lbuf);
            HB_STAMP(2)
            head_bx_layer<2, FXM, true>(smem_h + 2 * L_BYTES, bias_lds + 128, chain(accB, bias_v + 64), accA, lane, half, amax, lbuf);
            HB_STAMP(3)
            f32x16 lg[DUST ? 2 : 3];
            float lgd;                              // the dustbin logit
            if constexpr (DUST) {
                // the last layer's input passes through the lambda as fp32: the dustbin's dot product is taken there (the lane's 32 features; the other half-wave has
                // the other 32), its weights two float4 of the LDS table per K step like the biases
                float dust = 0.f;
                const float* bias_in = bias_v + 128;
                const float* dust_v = reinterpret_cast<const float*>(smem_h + bias_off + NB * 4);      // the same lane view (+ 4 half) of the dustbin weights behind the biases
                head_bx_layer<2, FXM, true>(smem_h + 3 * L_BYTES, bias_lds + 192, [&](int t, float (&y)[8]) {
                    const float4 b0 = *reinterpret_cast<const float4*>(bias_in + (t >> 1) * 32 + 16 * (t & 1)), b1 = *reinterpret_cast<const float4*>(bias_in + (t >> 1) * 32 + 16 * (t & 1) + 8);
                    const float4 d0 = *reinterpret_cast<const float4*>(dust_v + (t >> 1) * 32 + 16 * (t & 1)), d1 = *reinterpret_cast<const float4*>(dust_v + (t >> 1) * 32 + 16 * (t & 1) + 8);
                    const float bq[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w}, dq[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        y[i] = fmaxf(fmaf(accA[t >> 1][8 * (t & 1) + i], FX_SCALE_INV, bq[i]), 0.f);
                        dust = fmaf(y[i], dq[i], dust);
                    }
                }, lg, lane, half, amax, lbuf);
                lgd = dust + xhalf(dust) + a.b_dust;
            } else {
                head_bx_layer<3, FXM, true>(smem_h + 3 * L_BYTES, bias_lds + 192, chain(accA, bias_v + 128), lg, lane, half, amax, lbuf);
                lgd = lg[DUST ? 0 : 2][0];
            }
            if constexpr (FX) {      // logits = 2^-11 acc + bias
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const float4 bq = *reinterpret_cast<const float4*>(bias_v + 192 + m * 32 + 8 * g4);
                        lg[m][4 * g4] = fmaf(lg[m][4 * g4], FX_SCALE_INV, bq.x); lg[m][4 * g4 + 1] = fmaf(lg[m][4 * g4 + 1], FX_SCALE_INV, bq.y);
                        lg[m][4 * g4 + 2] = fmaf(lg[m][4 * g4 + 2], FX_SCALE_INV, bq.z); lg[m][4 * g4 + 3] = fmaf(lg[m][4 * g4 + 3], FX_SCALE_INV, bq.w);
                    }
            }
            HB_STAMP(4)
            if (LATE_X && tile + (int)gridDim.x < a.ntiles) issue_x(tile + gridDim.x);
            // lane (l31,half) holds logits c = 32m + (r&3) + 8(r>>2) + 4*half of its cell; c == 64 (dustbin) is m=2,r=0,half=0
            float mx = -INFINITY;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, lg[m][r]);
            if (half == 0) mx = fmaxf(mx, lgd);
            mx = fmaxf(mx, xhalf(mx));
            float sum = 0.f;
            f32x16 e[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) { e[m][r] = expf(lg[m][r] - mx); sum += e[m][r]; }
            if (half == 0) sum += expf(lgd - mx);
            sum += xhalf(sum);
            if (gcell < a.ncell) {
                const int b = gcell / hw, rem = gcell - b * hw;
                const int ci = rem / a.wc, cj = rem - ci * a.wc;
                float* o = a.out + (size_t)b * a.H * a.W + (size_t)(8 * ci) * a.W + 8 * cj + 4 * half;
                const float rs = 1.f / sum;                // one correctly-rounded divide, then 64 multiplies
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {          // dy = q + 4m, dx = 4*half .. +3
                        const float4 v = make_float4(e[m][4 * q] * rs, e[m][4 * q + 1] * rs, e[m][4 * q + 2] * rs, e[m][4 * q + 3] * rs);
                        *reinterpret_cast<float4*>(o + (size_t)(q + 4 * m) * a.W) = v;
                    }
                if (a.logits) {
                    float* lp = a.logits + (size_t)gcell * 65;
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) lp[m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = lg[m][r];
                    if (half == 0) lp[64] = lgd;
                }
            }
            HB_STAMP(5)
        } else {
            head_bx_layer<2, FXM, true>(smem_h + L_BYTES, bias_lds + 64, chain(accA, bias_v), accB, lane, half, amax, lbuf);
            // final 64 -> 1: dot over this lane's 32 channels, other half via one shuffle
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float v = FX ? fmaf(accB[m][r], FX_SCALE_INV, bias_v[64 + m * 32 + (r & 3) + 8 * (r >> 2)]) : accB[m][r];
                    s = fmaf(fmaxf(v, 0.f), a.w_last[ch], s);
                }
            s += __shfl_xor(s, 32, 64);
            if (half == 0 && gcell < a.ncell) a.out[gcell] = 1.f / (1.f + expf(-(s + a.b_last)));
        }
    }
    if constexpr (FX) fx_report_h(amax, a.status);
#undef HB_STAMP
}


}  // namespace xfh
