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
// the vector-ALU form (fp32's range): the fallback of the fp16 pair's range guard
__global__ __launch_bounds__(512) void block1_fused_kernel(XFH_B1_PARAMS) { block1_fused_body<5>(XFH_B1_ARGS); }
// the default: conv3, conv4 on the fp16 matrix cores.  Three workgroups per CU as the vector form: six waves per SIMD = at most 80 vector registers
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 8))) void block1_mx_kernel(XFH_B1_PARAMS) { block1_fused_body<7>(XFH_B1_ARGS); }
#undef XFH_B1_PARAMS
#undef XFH_B1_ARGS

// What was measured on the way (MI355X, B = 64 VGA; FINDINGS.md has the long form):
//   * v_pk_fma_f32 (broadcast A, SGPR-pair B, the form hipcc emits here) issues at the full packed rate; two v_fma_f32 doing the same work run at 0.6 of it:
//     the vector stages must stay SLP-packable (written as explicit 2-vectors).
//   * conv1 recomputed inside conv2 (no c1 tile: + 12 % FLOPs, one stage, one barrier and 26 KB of LDS less: three workgroups per CU instead of two): - 8 %.
//   * conv3 / conv4 on v_mfma_f32_16x16x4_f32 (round 3): slower -- an f32 MFMA has the FLOP rate of the packed vector FMA; on the fp16 matrix cores in the fp16-pair
//     arithmetic (round 5, this kernel): 212-264 -> 159-190 us.
//   * conv2 (4 -> 8, stride 2, K = 36) on the matrix cores, by instruction count: of stage 2's 306 v_pk_fma_f32 per c2 pixel 162 are conv1's
//     recomputation and 144 conv2's.  Without a c1 tile the B operand must be produced by the lane that owns it (a column = a pair of c2 pixels, K = 3 x 5 x 4 = 60 of
//     64): 15 c1 pixels per pair instead of 18, but every lane computes four of them with their own index arithmetic, 24 LDS reads, ReLU and fp16-pair split:
//     ~172 vector instructions per lane and block of 32 pixels = 344 per c2 pixel against today's ~400 -- stage 2 is 45 % of the kernel, so <= 7 % of the kernel
//     before the 21.4 blocks meet 8 waves (3 per wave: 0.89).  With a c1 tile (conv1 once per pixel) LDS grows by 44 KB to 80 KB: two workgroups per CU, the
//     configuration that measured 8 % slower than three.  Not built.

// variant: 7 = block1_mx_kernel (needs the fp16-pair weight images: a layer weight of magnitude >= kFxMaxWeight leaves them NULL), anything else = the vector form
void launch_block1_fused(const NetWeights& nw, const float* gray, const float* coef, int B, int H, int W, float* x1, hipStream_t st, int variant, int* status) {
    const ConvW& c0 = nw.conv[L_BLOCK1_0];
    const ConvW& c1w = nw.conv[L_BLOCK1_1];
    const ConvW& c2 = nw.conv[L_BLOCK1_2];
    const ConvW& c3 = nw.conv[L_BLOCK1_3];
    const ConvW& sk = nw.conv[L_SKIP1];
    const int H4 = H / 4, W4 = W / 4;
    const int tx = ceil_div(W4, b1::OW), ty = ceil_div(H4, b1::OH);
    static AttrMask attr5{0}, attr7{0};
    if (variant == 7 && nw.block1_fx && nw.block1_fx3) {      // (+ conv3's weight image behind the tiles: 53.6 KB, still three workgroups per CU)
        constexpr int lds7 = b1::M_LDS_FLOATS * 4 + b1fx::W3_BYTES;
        static_assert(3 * ((lds7 + 1279) / 1280 * 1280) <= 160 * 1024, "three workgroups per CU, also with 1280-byte allocation granules");
        set_max_dynamic_lds(reinterpret_cast<const void*>(block1_mx_kernel), lds7, attr7);
        block1_mx_kernel<<<xcd_grid_size(tx * ty, B), 512, lds7, st>>>(gray, coef, x1, B, H, W, tx, ty, c0.w_kc, c0.bias, c1w.w_kc, c1w.bias, c2.w_kc, c2.bias, c3.w_kc,
                                                                        c3.bias, sk.w_oihw, sk.bias, nw.block1_fx, nw.block1_fx3, status, g_debug_cold);
    } else {
        set_max_dynamic_lds(reinterpret_cast<const void*>(block1_fused_kernel), b1::F_LDS_FLOATS * 4, attr5);
        block1_fused_kernel<<<xcd_grid_size(tx * ty, B), 512, b1::F_LDS_FLOATS * 4, st>>>(gray, coef, x1, B, H, W, tx, ty, c0.w_kc, c0.bias, c1w.w_kc, c1w.bias, c2.w_kc, c2.bias, c3.w_kc,
                                                                                           c3.bias, sk.w_oihw, sk.bias, nw.block1_fx, nw.block1_fx3, status, g_debug_cold);
    }
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
