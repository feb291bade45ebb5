// The split-operand heads (head_bx_kernel, k_heads.hip): the kernel body in a header of its own so that tests/emu/ can compile the SAME source for the host
// (XFH_HOST_EMU) and run it against a float64 reference without a GPU.
#pragma once
#ifndef XFH_HOST_EMU
#include "kernels.hpp"
#include <type_traits>
#define XFH_DYN_LDS_BYTES(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
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
// The heads on the fp16 matrix cores in the fp16-pair arithmetic (bx_split.hpp: three v_mfma_f32_32x32x16_f16 per K = 16 carry an fp32 product sum).
//
// A head is a chain of K = 64 layers on a wave's 32 cells, so nothing is staged: the weight fragments of every layer sit in LDS in operand
// order (96 KiB for the key-point head), the activations stay in registers -- a layer's D registers (lane = cell, registers = features
// (r & 3) + 8 (r >> 2) + 4 half) become the next layer's B fragments once ReLU'd and split (K step t of lane half h takes the register
// quads 8 (t & 1) and 8 (t & 1) + 4 of block t >> 1: features 32 (t >> 1) + 16 (t & 1) + 8 q + 4 h + i; the weights are packed in that K
// order) -- and the first layer's eight consecutive channels per K step are 32 contiguous bytes of the source (one pixel row of the 8x8
// cell for the unfold, one slice of the channels-last feature row), loaded straight into registers one tile ahead.  No barrier after the
// weights have landed: the eight waves of a workgroup drift apart and fill each other's split / softmax phases.
// A layer's accumulators live at scale 2^11 without the bias; the consumer applies fma(acc, 2^-11, bias) (the chain in front of its ReLU, the epilogues on the logits).
// ------------------------------------------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct HeadBxArgs {
    const float* src;        // KP: raw gray (B,H,W) ; REL: feats (B*hc*wc, 64)
    const float* coef;       // KP: per-image instance-norm {alpha, beta}
    const uint4* wq;         // all layers: [layer][K step 4][cout block][fragment q0, q1, q2][64 lanes] 8 fp16   (weight_split.hpp: pack_head_layer)
    const float* bias;       // all layers' biases, padded to the cout blocks: KP 64,64,64,96 ; REL 64,64
    const float* w_last;     // REL: the 64 weights of the final 64 -> 1 layer
    float b_last;            //      and its bias
    float* out;              // KP: heat (B,H,W) ; REL: reliability (B*hc*wc)
    float* logits;           // KP only, optional: (B*hc*wc, 65)
    float* inv;              // REL only, optional: 1 / max(||feats[cell,:]||, 1e-12)
    int H, W, hc, wc, ncell, ntiles;
    long long* trace;        // debug: s_memtime stamps of wave 0's second tile, 16 per workgroup
    int cold;                // debug (xfh_debug_cold_start)
    int* status;             // range guard (bx_split.hpp), may be NULL
    const float* w_dust;     // KP: the 64 fp32 weights of the dustbin logit (row 64 of keypoint_head.3) -- computed as a dot product on the vector ALUs,
    float b_dust;            //     and its bias: the last layer then has two cout blocks instead of three (12 of 36 MFMAs and 16 accumulator registers less)
};

// one K = 64 layer: out[mb] = W x at scale 2^11 WITHOUT the bias (the consumer applies fma(out, 2^-11, bias)), x given per K step by `xs(t, y[8])`; weights of the layer at
// wl (LDS, operand order); amax collects the largest |x| converted (range guard)
template <int MBO, bool RELU_IN, typename XS>      // RELU_IN: xs delivers non-negative values (a chained layer): cheaper range tracking
__device__ inline void head_bx_layer(const unsigned char* wl, XS xs, f32x16 (&out)[MBO], int lane, int half, unsigned& amax) {
    constexpr int NW = 3;       // weight fragments per (K step, cout block) in LDS
    using frag_t = f16x8;
    constexpr int NXS = 2;      // input fragments (high parts, low parts)
#pragma unroll
    for (int mb = 0; mb < MBO; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[mb][r] = 0.f;
    // (compiler fence: the weights never change, so hipcc hoists every fragment read of every layer out of the persistent tile loop --
    // 432 registers' worth, straight into scratch memory)
    asm volatile("" ::: "memory");
    frag_t w[2][MBO][3];
    auto ldw = [&](int t, frag_t (&o)[MBO][3]) {
#pragma unroll
        for (int mb = 0; mb < MBO; ++mb)
#pragma unroll
            for (int q = 0; q < NW; ++q) o[mb][q] = *reinterpret_cast<const frag_t*>(wl + (((t * MBO + mb) * NW + q) * 64 + lane) * 16);
    };
    ldw(0, w[0]);
    // The B fragments are VALU results, and a VALU write that follows an MFMA by a few cycles can land in that MFMA's A/B registers
    // before the matrix core has read all 64 lanes of them (seen here: the cells of lanes 16-31 of a wave wrong in a few launches out
    // of many; hipcc's hazard recogniser only covers SrcC).  So the fragments of step t+1 are built into a second register set while
    // step t's are still alive: the empty asm below is a use of step t's set AFTER the split, which keeps the allocator from handing
    // its registers to the new values; a set is rewritten one full step (12-18 MFMAs) after its last read.
    frag_t xf[2][NXS];
    auto split8 = [&](const float (&y)[8], frag_t (&o)[NXS]) {
        uint4 uh, ul;
        unsigned* ph = &uh.x; unsigned* pl = &ul.x;
#pragma unroll
        for (int i = 0; i < 4; ++i) { split2_f16(y[2 * i], y[2 * i + 1], ph[i], pl[i]); fx_track_h(amax, ph[i], !RELU_IN); }
        o[0] = __builtin_bit_cast(frag_t, uh); o[1] = __builtin_bit_cast(frag_t, ul);
    };
    {
        float y[8];
        xs(0, y);
        split8(y, xf[0]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t + 1 < 4) ldw(t + 1, w[(t + 1) & 1]);
        asm volatile("" ::: "memory");
        const frag_t x0 = xf[t & 1][0], x1 = xf[t & 1][1];
        __builtin_amdgcn_sched_barrier(0);      // the MFMAs of a step stay together: left free, hipcc floats the NEXT steps' splits in between them
        // products (weight fragment, input fragment), small terms first: (q2, xh) (q1, xl) (q0, xh)
#define HB_MM(WQ, X) { _Pragma("unroll") for (int mb = 0; mb < MBO; ++mb) out[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[t & 1][mb][WQ], X, out[mb], 0, 0, 0); }
        HB_MM(2, x0) HB_MM(1, x1) HB_MM(0, x0)
#undef HB_MM
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < 4) {
            float y[8];
            xs(t + 1, y);
            split8(y, xf[(t + 1) & 1]);
            // the new fragments pass THROUGH the asm that uses the old ones (and this step's weights): it cannot move above the split,
            // so the old registers stay occupied while the split's results and temporaries are written
#ifndef XFH_HOST_EMU
            asm volatile("" : "+v"(xf[(t + 1) & 1][0]), "+v"(xf[(t + 1) & 1][1])
                            : "v"(xf[t & 1][0]), "v"(xf[t & 1][1]), "v"(w[t & 1][0][0]), "v"(w[t & 1][0][1]), "v"(w[t & 1][0][2]),
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

// The persistent head kernel's body.  KP: key-point head (8x8 unfold of the raw gray image with the instance normalisation applied on load -> 3 x [64 -> 64 + ReLU] ->
// 64 -> 64 logits on the matrix cores + the dustbin logit as a vector dot product -> softmax, depth-to-space); else the reliability head (feats -> 2 x [64 -> 64 + ReLU] -> 64 -> 1 ->
// sigmoid, + 1 / |feats|).  Weights a.wq = NetWeights::head_fx.
template <bool KP>
__device__ __forceinline__ void head_bx_body(const HeadBxArgs& a) {
    constexpr int NW = 3;
    constexpr int NB = KP ? 288 : 128;                                 // bias floats
    constexpr bool DUST = KP;                                         // the dustbin logit as a dot product (HeadBxArgs::w_dust)
    constexpr int W_BYTES = (KP ? 8 : 4) * 4 * NW * 1024;              // cout blocks x K steps x fragments x 1 KiB
    constexpr int L_BYTES = 2 * 4 * NW * 1024;                        // a 64 -> 64 layer
    XFH_DYN_LDS_BYTES(smem_h);
    float* bias_lds = reinterpret_cast<float*>(smem_h + W_BYTES);
    float* dust_lds = bias_lds + NB;                                   // DUST: the 64 dustbin weights
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
    unsigned amax = 0;                                                // the largest fp16 high parts converted, as a pair of 16-bit magnitudes (range guard: bx_split.hpp)
    // the lane's view of the bias table (+ 4 half), as ONE opaque 32-bit LDS offset plus small constants: left to itself hipcc gives every read of the table an
    // address register of its own (the table sits beyond the 64-KiB reach of an offset field), hoisted out of the tile loop and spilled; and the offset is pinned as an
    // INTEGER so that the pointer keeps its LDS type (a pinned pointer becomes generic: flat loads)
    int bias_off = W_BYTES + 16 * half;
    XFH_PIN(bias_off);
    const float* bias_v = reinterpret_cast<const float*>(smem_h + bias_off);                                  // an address register of its own (the table sits beyond the 64-KiB reach of an offset field), hoisted out of the tile loop and spilled
    long long* tr = a.trace && tid == 0 ? a.trace + (size_t)blockIdx.x * 16 : nullptr;
    int tix = 0;
#define HB_STAMP(k) { if (tr && tix == 1) tr[k] = __builtin_amdgcn_s_memtime(); }
    for (; tile < a.ntiles; tile += gridDim.x, ++tix) {
        const int gcell = tile * HD_CELLS + wave * 32 + l31;          // this lane's cell
        HB_STAMP(0)
        f32x16 accA[2], accB[2];
        float nrm2 = 0.f;
        {
            const float al = nalpha, be = nbeta;
            head_bx_layer<2, false>(smem_h, [&](int t, float (&y)[8]) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    y[i] = KP ? fmaf(xin[t][i], al, be) : xin[t][i];
                    if (!KP) nrm2 = fmaf(y[i], y[i], nrm2);
                }
            }, accA, lane, half, amax);
        }
        if (!KP && a.inv) {
            nrm2 += xhalf(nrm2);                                      // the other 32 channels sit in the other half-wave
            if (half == 0 && gcell < a.ncell) a.inv[gcell] = 1.f / fmaxf(sqrtf(nrm2), 1e-12f);
        }
        HB_STAMP(1)
        constexpr bool LATE_X = false;                // (a form without 32 registers to spare during the chained layers would prefetch under the softmax only: none needs it since the dustbin logit left the matrix cores)
        if (!LATE_X && tile + (int)gridDim.x < a.ntiles) issue_x(tile + gridDim.x);      // the next tile's input flies during the chained layers
        // chained layers: K step t = register quads 8 (t & 1), 8 (t & 1) + 4 of block t >> 1, ReLU'd
        // (`in` is at scale 2^11 and without its bias: both applied here -- the biases of the lane's eight features are two float4 of the LDS table)
        auto chain = [](const f32x16 (&in)[2], const float* bias_in) {      // bias_in: this lane's view of the table (+ 4 half)
            return [&in, bias_in](int t, float (&y)[8]) {
                const float4 b0 = *reinterpret_cast<const float4*>(bias_in + (t >> 1) * 32 + 16 * (t & 1));
                const float4 b1 = *reinterpret_cast<const float4*>(bias_in + (t >> 1) * 32 + 16 * (t & 1) + 8);
                const float bq[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int i = 0; i < 8; ++i) y[i] = fmaxf(fmaf(in[t >> 1][8 * (t & 1) + i], FX_SCALE_INV, bq[i]), 0.f);
            };
        };
        if (KP) {
            head_bx_layer<2, true>(smem_h + L_BYTES, chain(accA, bias_v), accB, lane, half, amax);
            HB_STAMP(2)
            head_bx_layer<2, true>(smem_h + 2 * L_BYTES, chain(accB, bias_v + 64), accA, lane, half, amax);
            HB_STAMP(3)
            f32x16 lg[2];
            float lgd;                              // the dustbin logit
            {
                // the last layer's input passes through the lambda as fp32: the dustbin's dot product is taken there (the lane's 32 features; the other half-wave has
                // the other 32), its weights two float4 of the LDS table per K step like the biases
                float dust = 0.f;
                const float* bias_in = bias_v + 128;
                const float* dust_v = reinterpret_cast<const float*>(smem_h + bias_off + NB * 4);      // the same lane view (+ 4 half) of the dustbin weights behind the biases
                head_bx_layer<2, true>(smem_h + 3 * L_BYTES, [&](int t, float (&y)[8]) {
                    const float4 b0 = *reinterpret_cast<const float4*>(bias_in + (t >> 1) * 32 + 16 * (t & 1)), b1 = *reinterpret_cast<const float4*>(bias_in + (t >> 1) * 32 + 16 * (t & 1) + 8);
                    const float4 d0 = *reinterpret_cast<const float4*>(dust_v + (t >> 1) * 32 + 16 * (t & 1)), d1 = *reinterpret_cast<const float4*>(dust_v + (t >> 1) * 32 + 16 * (t & 1) + 8);
                    const float bq[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w}, dq[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        y[i] = fmaxf(fmaf(accA[t >> 1][8 * (t & 1) + i], FX_SCALE_INV, bq[i]), 0.f);
                        dust = fmaf(y[i], dq[i], dust);
                    }
                }, lg, lane, half, amax);
                lgd = dust + xhalf(dust) + a.b_dust;
            }
            {      // logits = 2^-11 acc + bias
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
            head_bx_layer<2, true>(smem_h + L_BYTES, chain(accA, bias_v), accB, lane, half, amax);
            // final 64 -> 1: dot over this lane's 32 channels, other half via one shuffle
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float v = fmaf(accB[m][r], FX_SCALE_INV, bias_v[64 + m * 32 + (r & 3) + 8 * (r >> 2)]);
                    s = fmaf(fmaxf(v, 0.f), a.w_last[ch], s);
                }
            s += __shfl_xor(s, 32, 64);
            if (half == 0 && gcell < a.ncell) a.out[gcell] = 1.f / (1.f + expf(-(s + a.b_last)));
        }
    }
    fx_report_h(amax, a.status);
#undef HB_STAMP
}


}  // namespace xfh
