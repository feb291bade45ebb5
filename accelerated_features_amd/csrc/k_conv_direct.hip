// Direct (VALU) convolutions.
//
//  * conv3x3_small_kernel<CIN,COUT,STRIDE>: the low-channel, high-resolution layers of block1
//    (1->4, 4->8 s2, 8->8, 8->24 s2; modules/model.py:43-48).  These are HBM-bound (3.6-27
//    FLOP/B): one thread per output pixel keeps all COUT accumulators in registers, rows are
//    read coalesced (consecutive lanes = consecutive x), weights come through the scalar cache
//    (uniform addresses -> s_load, used as SGPR operands of v_fmac).  The last layer fuses
//    skip1 = AvgPool2d(4,4) + Conv2d(1,24,1)+bias and the residual add (model.py:40-41,140).
//  * conv_direct_generic_kernel: any layer, slow; the independent check used by
//    xfh_conv_layer(variant=1) to A/B the MFMA kernels on the device.
#include "kernels.hpp"
#include "bx_split.hpp"
#include "block1_fx.hpp"
#include <type_traits>
#include <cstdlib>
#include <cstring>

namespace xfh {

template <int CIN, int COUT, int STRIDE, bool SKIP>
__global__ __launch_bounds__(256) void conv3x3_small_kernel(const float* __restrict__ in, const float* __restrict__ wk,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int Hin, int Win, int Hout, int Wout,
                                                            const float* __restrict__ gray, int Hg, int Wg,
                                                            const float* __restrict__ skw, const float* __restrict__ skb) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z;
    if (ox >= Wout || oy >= Hout) return;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
    const float* inb = in + (size_t)b * CIN * Hin * Win;
    const int iy0 = oy * STRIDE - 1, ix0 = ox * STRIDE - 1;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
        const float* pl = inb + (size_t)ci * Hin * Win;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = iy0 + dy;
            const bool yok = (iy >= 0) && (iy < Hin);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ix0 + dx;
                float v = 0.f;
                if (yok && ix >= 0 && ix < Win) v = pl[iy * Win + ix];
                const float* w = wk + ((ci * 9) + dy * 3 + dx) * COUT;   // uniform -> scalar loads
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[co] = fmaf(v, w[co], acc[co]);
            }
        }
    }
    float sk = 0.f;
    if (SKIP) {   // 4x4 average of the normalised gray image at this output pixel (Hg = 4*Hout)
        const float* g = gray + (size_t)b * Hg * Wg + (size_t)(oy * 4) * Wg + ox * 4;
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 v = *reinterpret_cast<const float4*>(g + r * Wg);
            s += v.x; s += v.y; s += v.z; s += v.w;
        }
        sk = s * 0.0625f;
    }
    float* ob = out + (size_t)b * COUT * Hout * Wout + (size_t)oy * Wout + ox;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        float v = fmaxf(acc[co] + bias[co], 0.f);
        if (SKIP) v += fmaf(sk, skw[co], skb[co]);
        ob[(size_t)co * Hout * Wout] = v;
    }
}

template <int CIN, int COUT, int STRIDE, bool SKIP>
static void launch_small(const float* in, const float* wk, const float* bias, float* out, int B, int Hin, int Win,
                         const float* gray, const float* skw, const float* skb, hipStream_t st) {
    const int Hout = (Hin - 1) / STRIDE + 1, Wout = (Win - 1) / STRIDE + 1;
    conv3x3_small_kernel<CIN, COUT, STRIDE, SKIP><<<dim3(ceil_div(Wout, 64), ceil_div(Hout, 4), B), 256, 0, st>>>(
        in, wk, bias, out, Hin, Win, Hout, Wout, gray, Hout * 4, Wout * 4, skw, skb);
}

// block1: gray (B,1,H,W) -> t0 (B,4,H,W) -> t1 (B,8,H/2,W/2) -> t2 (same) -> x1 (B,24,H/4,W/4) [+ skip1]
void launch_block1(const NetWeights& nw, const float* gray, int B, int H, int W, float* t0, float* t1, float* t2,
                   float* x1, hipStream_t st) {
    const ConvW& c0 = nw.conv[L_BLOCK1_0];
    const ConvW& c1 = nw.conv[L_BLOCK1_1];
    const ConvW& c2 = nw.conv[L_BLOCK1_2];
    const ConvW& c3 = nw.conv[L_BLOCK1_3];
    const ConvW& sk = nw.conv[L_SKIP1];
    launch_small<1, 4, 1, false>(gray, c0.w_kc, c0.bias, t0, B, H, W, nullptr, nullptr, nullptr, st);
    launch_small<4, 8, 2, false>(t0, c1.w_kc, c1.bias, t1, B, H, W, nullptr, nullptr, nullptr, st);
    launch_small<8, 8, 1, false>(t1, c2.w_kc, c2.bias, t2, B, H / 2, W / 2, nullptr, nullptr, nullptr, st);
    launch_small<8, 24, 2, true>(t2, c3.w_kc, c3.bias, x1, B, H / 2, W / 2, gray, sk.w_oihw, sk.bias, st);
}

int launch_block1_layer(const NetWeights& nw, int layer, const float* in, int B, int Hin, int Win, float* out,
                        hipStream_t st) {
    const ConvW& c = nw.conv[layer];
    switch (layer) {
        case L_BLOCK1_0: launch_small<1, 4, 1, false>(in, c.w_kc, c.bias, out, B, Hin, Win, nullptr, nullptr, nullptr, st); return 0;
        case L_BLOCK1_1: launch_small<4, 8, 2, false>(in, c.w_kc, c.bias, out, B, Hin, Win, nullptr, nullptr, nullptr, st); return 0;
        case L_BLOCK1_2: launch_small<8, 8, 1, false>(in, c.w_kc, c.bias, out, B, Hin, Win, nullptr, nullptr, nullptr, st); return 0;
        case L_BLOCK1_3: launch_small<8, 24, 2, false>(in, c.w_kc, c.bias, out, B, Hin, Win, nullptr, nullptr, nullptr, st); return 0;
    }
    return -1;
}

}  // namespace xfh
#include "block1_body.hpp"      // block1_fused_body<MODE>: the fused block1 + skip1 (also compiled for the host by tests/emu/)
namespace xfh {

#define XFH_B1_PARAMS const float* __restrict__ gray, const float* __restrict__ coef, float* __restrict__ x1, int B, int H, int W, int tiles_x, int tiles_y,                      \
                      const float* __restrict__ w1, const float* __restrict__ bb1, const float* __restrict__ w2, const float* __restrict__ bb2, const float* __restrict__ w3,     \
                      const float* __restrict__ bb3, const float* __restrict__ w4, const float* __restrict__ bb4, const float* __restrict__ skw, const float* __restrict__ skb, \
                      const void* __restrict__ w4fx, const void* __restrict__ w3fx, int* __restrict__ status, int cold
#define XFH_B1_ARGS gray, coef, x1, B, H, W, tiles_x, tiles_y, w1, bb1, w2, bb2, w3, bb3, w4, bb4, skw, skb, w4fx, w3fx, status, cold
template <int C1MODE>
__global__ __launch_bounds__(512) void block1_fused_kernel(XFH_B1_PARAMS) { block1_fused_body<C1MODE>(XFH_B1_ARGS); }
// modes 6, 7 keep the three workgroups per CU of mode 5: six waves per SIMD = at most 80 vector registers
template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 8))) void block1_mx_kernel(XFH_B1_PARAMS) { block1_fused_body<MODE>(XFH_B1_ARGS); }
#undef XFH_B1_PARAMS
#undef XFH_B1_ARGS

// Split-bf16 MFMA variants of block1_fused_kernel (round 2: measured, identical results within fp32 rounding, slower, removed -- DESIGN 3.6).
// conv3 (8 -> 8) and conv4 (8 -> 24, s2) have K = 72 = 9 taps x 8 channels: four taps per v_mfma_f32_16x16x32_bf16, three K steps of six MFMAs;
// conv2's / conv3's epilogues wrote their 8 channels per pixel as three bf16 rows into LDS (48 B per pixel).
// A variant with ONLY conv4 on the matrix cores (same tiles and LDS budget as the kernel above, one tile per workgroup, nothing held in
// registers across stages) was no better: conv4 6.2 k -> 7.2 k cycles as written (its 36 MFMAs per wave are two dependent accumulator chains:
// ~150 cycles per v_mfma_f32_16x16x32_bf16 with the operand reads in between; four chains would bring it to ~3 k) while conv3's epilogue pays
// 2.6 k for splitting its outputs into bf16 rows -- a wash at best.  Removed.
// With conv3 AND conv4 there (persistent workgroups, conv3's weights in registers): per tile 5.3 k cycles gray + 7.4 k conv1 + 7.5 k conv2 (+ split) + 9-10 k conv3 + 4-10 k conv4 = 34-41 k against
// 32.9 k for the kernel above (325-371 us against 294).  conv3 / conv4 needed 90 + 36 MFMAs per wave (~2 k cycles of the pipe) but at 128 VGPRs
// and ~100 SGPRs (two workgroups of 8 waves per CU, the VALU stages' scalar weight streams, a persistent loop, 14 pointer arguments) hipcc
// spilled both register files, and every scratch reload parked a vmcnt(0) in the MFMA stages.  What it would take: conv3's weights in LDS
// (no room next to C2s + C3s + conv4's weights in 80 KB), or one 8-wave workgroup per CU with 256 registers and the two halves of the
// workgroup a stage apart.
// Measured on MI355X, B = 64 VGA (round 2, tools/bench_src/pk_fma_chain.hip + in-kernel s_memtime stamps):
//   * v_pk_fma_f32 (broadcast A, SGPR-pair B, the form hipcc emits here) issues at the full packed rate (115-123 TFLOP/s chip-wide
//     at 8 or 16 waves per CU); two v_fma_f32 doing the same work run at 70: everything below must stay SLP-packable.
//   * this kernel: 292 us = 48 TFLOP/s.  With the weight loads AND the LDS reads made loop-invariant (hoisted) it still takes
//     258 us: neither scalar-cache latency nor LDS conflicts bound it.  A persistent variant (next tile's gray prefetched into
//     registers, column-parity de-interleaved c1/c3 tiles = no bank conflicts, two columns per thread in conv1) measured 343 us
//     with bit-identical results and was dropped: per tile 1.1 k cycles stage 0, 8.6 k conv1, 6.1 k conv2, 9.9 k conv3, 6.2 k conv4,
//     2.4 k barriers -- the SIMDs issue FMAs ~45 % of the time; the rest is the lock-step stage structure (9 and 11 wave-loads
//     on 8 waves, two barriers per stage, two workgroups per CU to cover each other).  Unrolling the channel loops made it slower
//     (x2: +4 %, x8: +20 %).  What helped: issuing the six gray loads of a thread together (311 -> 292 us).
//   * 7 x 16 output tiles (the conv3 tile = 15 x 33 = 495 pixels = ONE pass of the 512 threads instead of 561 = a full pass + one wave,
//     conv1 5 passes instead of 6: 11 % fewer issue slots on a workgroup's critical path, 5 % more tiles): 299 us against 288 in
//     three alternating in-run pairs -- the partial passes are not what the stages wait for; dropped.
//   * workgroup size (same tiles, conv4 on 24 / (threads / 128) couts per thread): 256 threads 330 us, 512 threads 280 us, 1024 threads
//     380 us (alternating in-run pairs): with 16 waves conv4 reads its 72 inputs twice as often per FMA, with 4 waves nothing hides the LDS latency.

void launch_block1_fused(const NetWeights& nw, const float* gray, const float* coef, int B, int H, int W, float* x1, hipStream_t st, int variant, int* status) {
    const ConvW& c0 = nw.conv[L_BLOCK1_0];
    const ConvW& c1w = nw.conv[L_BLOCK1_1];
    const ConvW& c2 = nw.conv[L_BLOCK1_2];
    const ConvW& c3 = nw.conv[L_BLOCK1_3];
    const ConvW& sk = nw.conv[L_SKIP1];
    const int H4 = H / 4, W4 = W / 4;
    const int tx = ceil_div(W4, b1::OW), ty = ceil_div(H4, b1::OH);
    // conv1 on three adjacent pixels per thread: 275 -> 266 us in alternating in-run pairs (PMC: the kernel issues VALU instructions 76 % of the
    // time and only 54 % of them are FMAs -- index arithmetic, bounds and LDS addresses are the rest, and conv1 has the fewest FMAs per index).
    // Mode 4 writes the same three pixels as 2-vectors so that hipcc emits 54 v_pk_fma_f32 per item instead of 108 v_fmac_f32 (-108 of ~1400 VALU
    // instructions per thread); rocprof 280.5 us against 289.4 us for mode 3 on two comparable boxes in round 2.
    // Mode 5 (the default since round 3) drops the c1 tile altogether: conv2 recomputes its nine c1 pixels from 25 gray values in registers.
    // + 12 % FLOPs, but one stage, one barrier and 26 KB of LDS less: three workgroups per CU instead of two.  In-run A/B (tools/ab_option.py,
    // alternating rounds on one box): 264.5 / 262.9 / 262.3 -> 244.8 / 241.8 / 242.7 us, step 1.788 -> 1.753 ms.
    //
    // Round 3, measured and removed: conv3 + conv4 (70 % of the FLOPs, 5.8 k of the kernel's 12.1 k vector wave-instructions per tile) on
    // v_mfma_f32_16x16x4_f32 -- conv3 as N = 16 = two adjacent pixels x 8 couts over the union of their windows (K = 8 x 3 x 4 = 96, 72 used),
    // conv4 as two 16-wide cout blocks with K = 72; every A operand one ds_read_b32 at base(lane) + constant(step), B operands packed per step
    // and lane by the host, results back through LDS.  Parity-green on the first run (backbone / census tests), and 330 us against 283: an f32 MFMA
    // has the FLOP rate of the packed vector FMA, so the matrix stages take the cycles the vector stages took (6.0 k vs 5.8 k per SIMD and tile, 19
    // blocks on 8 waves), the two workgroups of a CU run their stages in phase (no matrix / vector overlap to collect), and the extra barrier
    // and LDS round trip come on top.  What the vector pipe is short of is issue slots (PMC: 116 M vector instructions, 54 % FMAs, inner loops
    // already 95 % v_pk_fma_f32) -- the remaining lever is the per-item prologue / epilogue arithmetic of the stages, not another pipe.
    const int c1 = (variant == 1 || variant == 3 || variant == 4) ? variant : ((variant == 6 || variant == 7) && nw.block1_fx && nw.block1_fx3) ? variant : 5;      // (6 / 7 without the fp16-pair images -- a weight of magnitude >= 31 -- are 5)      // option "block1": 1 = one pixel per thread; 3 = three pixels, scalar FMAs; 4 = three pixels on packed FMAs; default (0 / 5): conv1 recomputed inside conv2 (xfh_set_option rejects every other value)
    static AttrMask attr1{0}, attr3{0}, attr4{0}, attr5{0}, attr6{0}, attr7{0};
#define XFH_B1_LAUNCH(MODE, ATTR)                                                                                                       \
    {                                                                                                                                    \
        constexpr int lds_floats = MODE >= 5 ? b1::F_LDS_FLOATS : b1::LDS_FLOATS;                                                        \
        set_max_dynamic_lds(reinterpret_cast<const void*>(block1_fused_kernel<MODE>), lds_floats * 4, ATTR);                             \
        block1_fused_kernel<MODE><<<xcd_grid_size(tx * ty, B), 512, lds_floats * 4, st>>>(                                               \
            gray, coef, x1, B, H, W, tx, ty, c0.w_kc, c0.bias, c1w.w_kc, c1w.bias, c2.w_kc, c2.bias, c3.w_kc, c3.bias, sk.w_oihw, sk.bias, nw.block1_fx, nw.block1_fx3, status, g_debug_cold); \
    }
    if (c1 == 1) XFH_B1_LAUNCH(1, attr1)
    else if (c1 == 3) XFH_B1_LAUNCH(3, attr3)
    else if (c1 == 5) XFH_B1_LAUNCH(5, attr5)
    else if (c1 == 6) {
        set_max_dynamic_lds(reinterpret_cast<const void*>(block1_mx_kernel<6>), b1::M_LDS_FLOATS * 4, attr6);
        block1_mx_kernel<6><<<xcd_grid_size(tx * ty, B), 512, b1::M_LDS_FLOATS * 4, st>>>(gray, coef, x1, B, H, W, tx, ty, c0.w_kc, c0.bias, c1w.w_kc, c1w.bias, c2.w_kc, c2.bias, c3.w_kc,
                                                                                           c3.bias, sk.w_oihw, sk.bias, nw.block1_fx, nw.block1_fx3, status, g_debug_cold);
    } else if (c1 == 7) {      // (+ conv3's weight image behind the tiles: 53.6 KB, still three workgroups per CU)
        constexpr int lds7 = b1::M_LDS_FLOATS * 4 + b1fx::W3_BYTES;
        static_assert(3 * ((lds7 + 1279) / 1280 * 1280) <= 160 * 1024, "three workgroups per CU, also with 1280-byte allocation granules");
        set_max_dynamic_lds(reinterpret_cast<const void*>(block1_mx_kernel<7>), lds7, attr7);
        block1_mx_kernel<7><<<xcd_grid_size(tx * ty, B), 512, lds7, st>>>(gray, coef, x1, B, H, W, tx, ty, c0.w_kc, c0.bias, c1w.w_kc, c1w.bias, c2.w_kc, c2.bias, c3.w_kc,
                                                                           c3.bias, sk.w_oihw, sk.bias, nw.block1_fx, nw.block1_fx3, status, g_debug_cold);
    }
    else XFH_B1_LAUNCH(4, attr4)
#undef XFH_B1_LAUNCH
}

// ------------------------------------------------------------------------------------------
// generic direct conv: thread = (pixel, 8 output channels); weights in (Cout,Cin,k,k) order
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_direct_generic_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ out,
                                                                  int CIN, int COUT, int KS, int STRIDE, int relu,
                                                                  int Hin, int Win, int Hout, int Wout, int cogroups) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int b = blockIdx.z / cogroups, cg = blockIdx.z % cogroups;
    if (ox >= Wout || oy >= Hout) return;
    const int pad = KS / 2, KK = KS * KS;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const float* inb = in + (size_t)b * CIN * Hin * Win;
    for (int ci = 0; ci < CIN; ++ci) {
        const float* pl = inb + (size_t)ci * Hin * Win;
        for (int dy = 0; dy < KS; ++dy) {
            const int iy = oy * STRIDE - pad + dy;
            for (int dx = 0; dx < KS; ++dx) {
                const int ix = ox * STRIDE - pad + dx;
                float v = 0.f;
                if (iy >= 0 && iy < Hin && ix >= 0 && ix < Win) v = pl[iy * Win + ix];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int co = cg * 8 + j;
                    if (co < COUT) acc[j] = fmaf(v, w[((size_t)co * CIN + ci) * KK + dy * KS + dx], acc[j]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int co = cg * 8 + j;
        if (co < COUT) {
            float v = acc[j] + bias[co];
            if (relu) v = fmaxf(v, 0.f);
            out[(((size_t)b * COUT + co) * Hout + oy) * Wout + ox] = v;
        }
    }
}

void launch_conv_generic(const ConvW& c, const float* in, int B, int Hin, int Win, float* out, hipStream_t st) {
    const int pad = c.ks / 2;
    const int Hout = (Hin + 2 * pad - c.ks) / c.stride + 1, Wout = (Win + 2 * pad - c.ks) / c.stride + 1;
    const int cg = ceil_div(c.cout, 8);
    conv_direct_generic_kernel<<<dim3(ceil_div(Wout, 64), ceil_div(Hout, 4), B * cg), 256, 0, st>>>(
        in, c.w_oihw, c.bias, out, c.cin, c.cout, c.ks, c.stride, c.relu, Hin, Win, Hout, Wout, cg);
}

}  // namespace xfh
