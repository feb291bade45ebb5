// 3x3 / 1x1 convolution as an implicit GEMM on the f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Why MFMA here: the >=24-channel convolutions of XFeat (block2..block5, block_fusion;
// modules/model.py:50-77) carry 2.1 of the network's 2.6 GFLOP/frame at 42-144 FLOP/B, i.e.
// they are fp32-FMA bound, not HBM bound.  The f32-input MFMA is bit-for-bit an fp32 fma
// chain (no reduced precision) at the same 157 TFLOP/s peak as the vector ALU, but reaches
// that peak with one VGPR per operand and leaves the VALU free for address math.
//
// GEMM view (per image):  D[cout][pixel] = sum_{ci,tap} W[cout][ci,tap] * X[ci][pixel+tap]
//   A operand = weights  A[i=cout][k]   (lane l supplies i = l&31, k = l>>5)
//   B operand = input    B[k][j=pixel]  (lane l supplies j = l&31, k = l>>5)
//   one MFMA consumes the k-pair {(ci,tap),(ci+1,tap)}: lanes 0-31 take ci, lanes 32-63 ci+1.
//   D: lane holds pixel j = l&31 and couts (r&3)+8*(r>>2)+4*(l>>5), r = 0..15  -> NCHW
//   stores are 32 consecutive pixels per half-wave.  Swapping the operands transposes D,
//   which gives channels-last (NHWC) stores of 32 consecutive channels (used for M1).
//
// Work decomposition: one workgroup (4 waves) = TH x TW output pixels of one image x ALL output
// channels.  Each wave owns MB cout-blocks x NB pixel-blocks of 32x32 (MB*NB = 4 accumulators
// = 64 VGPRs).  K is walked in chunks of CK input channels: the chunk's weights
// [CK*k*k][COUT_PAD] and the input tile with halo [CK][IHt*IWt] are staged in LDS; the next
// chunk is prefetched into registers while the current one feeds the MFMAs.
// LDS reads are ds_read_b32 of 32 consecutive floats per half-wave: conflict-free for the
// weights, <=2-way for the pixels, and at one MFMA per 64 cycles LDS has >4x headroom.
#include "kernels.hpp"

namespace xfh {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvGeom {
    int Hin, Win, Hout, Wout;
    int TW, TH, tiles_x;
    int IWt, plane;
};

template <int CIN, int COUT, int KS, int STRIDE, int NPL, bool NHWC>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const float* __restrict__ in, const float* __restrict__ wk,
                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                        ConvGeom g, int relu) {
    constexpr int COUT_PAD = (COUT + 31) / 32 * 32;
    constexpr int MB = COUT_PAD / 32;
    constexpr int NB = MB == 1 ? 4 : (MB == 2 ? 2 : 1);
    constexpr int KK = KS * KS;
    constexpr int PAD = KS / 2;
    constexpr int CK = KS == 3 ? 8 : 32;
    constexpr int NCH = CIN / CK;
    static_assert(CIN % CK == 0, "CIN must be a multiple of the channel chunk");
    constexpr int WCH = CK * KK * COUT_PAD;          // floats per weight chunk
    constexpr int WV = (WCH / 4 + 255) / 256;        // float4 per thread per chunk

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;
    float* Xl = smem + WCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y;
    const int tyi = blockIdx.x / g.tiles_x, txi = blockIdx.x % g.tiles_x;
    const int oy0 = tyi * g.TH, ox0 = txi * g.TW;
    const int npix = g.TH * g.TW;
    const int plane = g.plane;
    const size_t HWin = (size_t)g.Hin * g.Win;
    const float* inb = in + (size_t)b * CIN * HWin;

    // where this thread's staged input elements come from (same for every channel)
    int goff[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int e = tid + i * 256;
        goff[i] = -1;
        if (e < plane) {
            const int r = e / g.IWt, c = e - r * g.IWt;
            const int gy = oy0 * STRIDE - PAD + r, gx = ox0 * STRIDE - PAD + c;
            if (gy >= 0 && gy < g.Hin && gx >= 0 && gx < g.Win) goff[i] = gy * g.Win + gx;
        }
    }
    // LDS offset of this lane's pixel for each of its pixel blocks (k-half folded in)
    int pixoff[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        int t = (wave * NB + n) * 32 + l31;
        if (t >= npix) t = 0;
        const int ty = t / g.TW, tx = t - ty * g.TW;
        pixoff[n] = half * plane + ty * STRIDE * g.IWt + tx * STRIDE;
    }
    const int wbase = half * KK * COUT_PAD + l31;

    f32x16 acc[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    float4 wreg[WV];
    float xreg[CK][NPL];

    auto prefetch = [&](int ch) {
        const float4* wsrc = reinterpret_cast<const float4*>(wk + (size_t)ch * WCH);
#pragma unroll
        for (int v = 0; v < WV; ++v) {
            const int e = tid + v * 256;
            if (e < WCH / 4) wreg[v] = wsrc[e];
        }
        const float* src = inb + (size_t)ch * CK * HWin;
#pragma unroll
        for (int c = 0; c < CK; ++c)
#pragma unroll
            for (int i = 0; i < NPL; ++i) xreg[c][i] = goff[i] >= 0 ? src[(size_t)c * HWin + goff[i]] : 0.f;
    };

    prefetch(0);
    for (int ch = 0; ch < NCH; ++ch) {
        __syncthreads();   // everyone is done reading the previous chunk
#pragma unroll
        for (int v = 0; v < WV; ++v) {
            const int e = tid + v * 256;
            if (e < WCH / 4) reinterpret_cast<float4*>(Wl)[e] = wreg[v];
        }
#pragma unroll
        for (int c = 0; c < CK; ++c)
#pragma unroll
            for (int i = 0; i < NPL; ++i) {
                const int e = tid + i * 256;
                if (e < plane) Xl[c * plane + e] = xreg[c][i];
            }
        __syncthreads();
        if (ch + 1 < NCH) prefetch(ch + 1);   // global loads fly while the MFMAs run

#pragma unroll
        for (int p = 0; p < CK / 2; ++p) {
#pragma unroll
            for (int tap = 0; tap < KK; ++tap) {
                const int toff = KS == 3 ? (tap / 3) * g.IWt + (tap % 3) : 0;
                float a[MB], bb[NB];
#pragma unroll
                for (int m = 0; m < MB; ++m) a[m] = Wl[wbase + ((2 * p) * KK + tap) * COUT_PAD + m * 32];
#pragma unroll
                for (int n = 0; n < NB; ++n) bb[n] = Xl[pixoff[n] + 2 * p * plane + toff];
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int n = 0; n < NB; ++n)
                        acc[m][n] = NHWC ? __builtin_amdgcn_mfma_f32_32x32x2f32(bb[n], a[m], acc[m][n], 0, 0, 0)
                                         : __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bb[n], acc[m][n], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: + folded-BN shift / bias, ReLU, store --------------------------------
    if (!NHWC) {
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int t = (wave * NB + n) * 32 + l31;
            const int ty = t / g.TW, tx = t - ty * g.TW;
            const int oy = oy0 + ty, ox = ox0 + tx;
            const bool ok = t < npix && oy < g.Hout && ox < g.Wout;
            float* op = out + ((size_t)b * COUT * g.Hout + oy) * g.Wout + ox;
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (ok && co < COUT) {
                        float v = acc[m][n][r] + bias[co];
                        if (relu) v = fmaxf(v, 0.f);
                        op[(size_t)co * g.Hout * g.Wout] = v;
                    }
                }
        }
    } else {
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = (wave * NB + n) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int ty = t / g.TW, tx = t - ty * g.TW;
                const int oy = oy0 + ty, ox = ox0 + tx;
                const bool ok = t < npix && oy < g.Hout && ox < g.Wout;
                float* op = out + (((size_t)b * g.Hout + oy) * g.Wout + ox) * COUT;
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const int co = m * 32 + l31;
                    if (ok && co < COUT) {
                        float v = acc[m][n][r] + bias[co];
                        if (relu) v = fmaxf(v, 0.f);
                        op[co] = v;
                    }
                }
            }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static bool choose_tile(int Hout, int Wout, int tile_pix, int S, int KS, int npl, ConvGeom& g) {
    long best_tiles = -1, best_cost = 0;
    for (int tw = 1; tw <= Wout && tw <= tile_pix; ++tw) {
        int th = tile_pix / tw;
        if (th > Hout) th = Hout;
        if (th < 1) continue;
        const int iwt = (tw - 1) * S + KS, iht = (th - 1) * S + KS;
        const long plane = (long)iwt * iht;
        if (plane > (long)npl * 256) continue;
        const long tiles = (long)ceil_div(Hout, th) * ceil_div(Wout, tw);
        const long cost = tiles * plane;
        if (best_tiles < 0 || tiles < best_tiles || (tiles == best_tiles && cost < best_cost)) {
            best_tiles = tiles; best_cost = cost;
            g.TW = tw; g.TH = th; g.IWt = iwt; g.plane = (int)plane;
        }
    }
    if (best_tiles < 0) return false;
    g.tiles_x = ceil_div(Wout, g.TW);
    return true;
}

template <int CIN, int COUT, int KS, int STRIDE, int NPL>
static int run(const ConvW& c, const float* in, int B, int Hin, int Win, float* out, bool nhwc, hipStream_t st) {
    constexpr int COUT_PAD = (COUT + 31) / 32 * 32;
    constexpr int MB = COUT_PAD / 32;
    constexpr int NB = MB == 1 ? 4 : (MB == 2 ? 2 : 1);
    constexpr int CK = KS == 3 ? 8 : 32;
    constexpr int WCH = CK * KS * KS * COUT_PAD;
    ConvGeom g;
    g.Hin = Hin; g.Win = Win;
    g.Hout = (Hin + 2 * (KS / 2) - KS) / STRIDE + 1;
    g.Wout = (Win + 2 * (KS / 2) - KS) / STRIDE + 1;
    if (!choose_tile(g.Hout, g.Wout, 4 * NB * 32, STRIDE, KS, NPL, g)) return -1;
    const int tiles = g.tiles_x * ceil_div(g.Hout, g.TH);
    const size_t lds = (size_t)(WCH + CK * g.plane) * sizeof(float);
    if (nhwc)
        conv_mfma_kernel<CIN, COUT, KS, STRIDE, NPL, true><<<dim3(tiles, B), 256, lds, st>>>(in, c.w_kcp, c.bias, out, g, c.relu);
    else
        conv_mfma_kernel<CIN, COUT, KS, STRIDE, NPL, false><<<dim3(tiles, B), 256, lds, st>>>(in, c.w_kcp, c.bias, out, g, c.relu);
    return 0;
}

int launch_conv_mfma(const ConvW& c, const float* in, int B, int Hin, int Win, float* out, bool nhwc, hipStream_t st) {
    const int key = c.cin * 1000000 + c.cout * 1000 + c.ks * 10 + c.stride;
    switch (key) {
        case 24 * 1000000 + 24 * 1000 + 31:   return run<24, 24, 3, 1, 3>(c, in, B, Hin, Win, out, nhwc, st);
        case 24 * 1000000 + 64 * 1000 + 32:   return run<24, 64, 3, 2, 5>(c, in, B, Hin, Win, out, nhwc, st);
        case 64 * 1000000 + 64 * 1000 + 31:   return run<64, 64, 3, 1, 2>(c, in, B, Hin, Win, out, nhwc, st);
        case 64 * 1000000 + 64 * 1000 + 32:   return run<64, 64, 3, 2, 5>(c, in, B, Hin, Win, out, nhwc, st);
        case 64 * 1000000 + 128 * 1000 + 32:  return run<64, 128, 3, 2, 3>(c, in, B, Hin, Win, out, nhwc, st);
        case 128 * 1000000 + 128 * 1000 + 31: return run<128, 128, 3, 1, 1>(c, in, B, Hin, Win, out, nhwc, st);
        case 64 * 1000000 + 64 * 1000 + 11:   return run<64, 64, 1, 1, 1>(c, in, B, Hin, Win, out, nhwc, st);
        case 128 * 1000000 + 64 * 1000 + 11:  return run<128, 64, 1, 1, 1>(c, in, B, Hin, Win, out, nhwc, st);
    }
    return -1;
}

double conv_flops(const ConvW& c, int B, int Hout, int Wout) {
    return 2.0 * B * Hout * Wout * (double)c.cout * c.cin * c.ks * c.ks;
}

}  // namespace xfh
