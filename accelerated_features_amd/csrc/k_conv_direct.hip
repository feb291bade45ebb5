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
