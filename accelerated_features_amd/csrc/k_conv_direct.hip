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

// ------------------------------------------------------------------------------------------
// block1 fused: gray (B,1,H,W) -> x1 = block1(gray) + skip1(gray)  (B,24,H/4,W/4)
//   (modules/model.py:40-48,140).  One workgroup = 8 x 16 output pixels.  The four
//   low-channel layers run back to back on LDS-resident tiles (halo recomputed per tile,
//   ~15 % extra FMAs), so the 4/8/8-channel full- and half-resolution activations
//   (19.7 MB/frame written and re-read by the layer-at-a-time version) never reach HBM:
//   the kernel reads the gray tile once and writes x1 once.
//
//   tile extents (rows x cols), origin in its own map:
//     out  8 x 16  at (Y4, X4)            [H/4 x W/4]
//     c3  17 x 33  at (2Y4-1, 2X4-1)      [H/2 x W/2]   conv3 8->8 s1
//     c2  19 x 35  at (2Y4-2, 2X4-2)      [H/2 x W/2]   conv2 4->8 s2
//     c1  39 x 71  at (4Y4-5, 4X4-5)      [H x W]       conv1 1->4 s1
//     g   41 x 73  at (4Y4-6, 4X4-6)      [H x W]       normalised gray
//   Positions outside a map are stored as 0 = the next conv's zero padding.
//   Weights are read with wave-uniform addresses (scalar loads, SGPR operands of v_fmac).
// ------------------------------------------------------------------------------------------
namespace b1 {
constexpr int OH = 8, OW = 16;
constexpr int C3H = 17, C3W = 33, C2H = 19, C2W = 35, C1H = 39, C1W = 71, GH = 41, GW = 73;
constexpr int G_OFF = 0, G_SZ = GH * GW;                  // 2993
constexpr int C1_OFF = G_OFF + G_SZ, C1_SZ = 4 * C1H * C1W;  // 11076
constexpr int C2_OFF = C1_OFF + C1_SZ, C2_SZ = 8 * C2H * C2W;  // 5320
constexpr int C3_OFF = C1_OFF;                            // overlays c1 (dead once c2 exists)
constexpr int LDS_FLOATS = C2_OFF + C2_SZ;                // 19389 floats = 77.6 KB
}  // namespace b1

__global__ __launch_bounds__(512) void block1_fused_kernel(const float* __restrict__ gray, const float* __restrict__ coef, float* __restrict__ x1, int B, int H, int W,
                                                           int tiles_x, int tiles_y,
                                                           const float* __restrict__ w1, const float* __restrict__ bb1,
                                                           const float* __restrict__ w2, const float* __restrict__ bb2,
                                                           const float* __restrict__ w3, const float* __restrict__ bb3,
                                                           const float* __restrict__ w4, const float* __restrict__ bb4,
                                                           const float* __restrict__ skw, const float* __restrict__ skb) {
    using namespace b1;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* G = lds + G_OFF;
    float* C1 = lds + C1_OFF;
    float* C2 = lds + C2_OFF;
    float* C3 = lds + C3_OFF;
    const int tid = threadIdx.x;
    // the tiles of an image run on one XCD: the 4-pixel gray halos of neighbouring tiles hit its L2
    int b, item;
    if (!xcd_group_map(blockIdx.x, tiles_x * tiles_y, B, b, item)) return;
    const int Y4 = (item / tiles_x) * OH, X4 = (item % tiles_x) * OW;
    const int H2 = H >> 1, W2 = W >> 1, H4 = H >> 2, W4 = W >> 2;
    const float* gb = gray + (size_t)b * H * W;

    // ---- stage 0: gray tile, instance-normalised on the way in (zero padding stays zero) -------
    const float alpha = coef[2 * b], beta = coef[2 * b + 1];
    {   // all six loads of a thread in flight together (as a rolled loop hipcc waits for each one: six exposed round trips per tile)
        constexpr int NL = (G_SZ + 511) / 512;
        float raw[NL];
        bool in[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int e = tid + k * 512;
            const int r = e / GW, c = e - r * GW;
            const int gy = 4 * Y4 - 6 + r, gx = 4 * X4 - 6 + c;
            in[k] = e < G_SZ && gy >= 0 && gy < H && gx >= 0 && gx < W;
            raw[k] = in[k] ? gb[(size_t)gy * W + gx] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int e = tid + k * 512;
            if (e < G_SZ) G[e] = in[k] ? fmaf(raw[k], alpha, beta) : 0.f;
        }
    }
    __syncthreads();

    // ---- stage 1: conv1 1->4, s1 --------------------------------------------------------------
    for (int e = tid; e < C1H * C1W; e += 512) {
        const int r = e / C1W, c = e - r * C1W;
        const int gy = 4 * Y4 - 5 + r, gx = 4 * X4 - 5 + c;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
#pragma unroll
            for (int co = 0; co < 4; ++co) acc[co] = bb1[co];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float v = G[(r + dy) * GW + c + dx];
                    const float* w = w1 + (dy * 3 + dx) * 4;
#pragma unroll
                    for (int co = 0; co < 4; ++co) acc[co] = fmaf(v, w[co], acc[co]);
                }
#pragma unroll
            for (int co = 0; co < 4; ++co) acc[co] = fmaxf(acc[co], 0.f);
        }
#pragma unroll
        for (int co = 0; co < 4; ++co) C1[co * (C1H * C1W) + e] = acc[co];
    }
    __syncthreads();

    // ---- stage 2: conv2 4->8, s2 --------------------------------------------------------------
    for (int e = tid; e < C2H * C2W; e += 512) {
        const int r = e / C2W, c = e - r * C2W;
        const int gy = 2 * Y4 - 2 + r, gx = 2 * X4 - 2 + c;
        float acc[8];
#pragma unroll
        for (int co = 0; co < 8; ++co) acc[co] = 0.f;
        if (gy >= 0 && gy < H2 && gx >= 0 && gx < W2) {
#pragma unroll
            for (int co = 0; co < 8; ++co) acc[co] = bb2[co];
#pragma unroll 1
            for (int ci = 0; ci < 4; ++ci) {
                const float* src = C1 + ci * (C1H * C1W) + (2 * r) * C1W + 2 * c;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float v = src[dy * C1W + dx];
                        const float* w = w2 + ((ci * 9) + dy * 3 + dx) * 8;
#pragma unroll
                        for (int co = 0; co < 8; ++co) acc[co] = fmaf(v, w[co], acc[co]);
                    }
            }
#pragma unroll
            for (int co = 0; co < 8; ++co) acc[co] = fmaxf(acc[co], 0.f);
        }
#pragma unroll
        for (int co = 0; co < 8; ++co) C2[co * (C2H * C2W) + e] = acc[co];
    }
    __syncthreads();

    // ---- stage 3: conv3 8->8, s1 (writes over the dead c1 tile) ----------------------------------
    for (int e = tid; e < C3H * C3W; e += 512) {
        const int r = e / C3W, c = e - r * C3W;
        const int gy = 2 * Y4 - 1 + r, gx = 2 * X4 - 1 + c;
        float acc[8];
#pragma unroll
        for (int co = 0; co < 8; ++co) acc[co] = 0.f;
        if (gy >= 0 && gy < H2 && gx >= 0 && gx < W2) {
#pragma unroll
            for (int co = 0; co < 8; ++co) acc[co] = bb3[co];
#pragma unroll 1
            for (int ci = 0; ci < 8; ++ci) {
                const float* src = C2 + ci * (C2H * C2W) + r * C2W + c;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float v = src[dy * C2W + dx];
                        const float* w = w3 + ((ci * 9) + dy * 3 + dx) * 8;
#pragma unroll
                        for (int co = 0; co < 8; ++co) acc[co] = fmaf(v, w[co], acc[co]);
                    }
            }
#pragma unroll
            for (int co = 0; co < 8; ++co) acc[co] = fmaxf(acc[co], 0.f);
        }
#pragma unroll
        for (int co = 0; co < 8; ++co) C3[co * (C3H * C3W) + e] = acc[co];
    }
    __syncthreads();

    // ---- stage 4: conv4 8->24, s2 + skip1 + residual add; thread = (pixel, 6 of 24 couts) -------
    {
        const int p = tid & 127, r = p >> 4, c = p & 15;
        const int g = __builtin_amdgcn_readfirstlane(tid >> 7);      // wave pair -> couts 6g .. 6g+5
        const int oy = Y4 + r, ox = X4 + c;
        float acc[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[j] = bb4[g * 6 + j];
#pragma unroll 1
        for (int ci = 0; ci < 8; ++ci) {
            const float* src = C3 + ci * (C3H * C3W) + (2 * r) * C3W + 2 * c;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float v = src[dy * C3W + dx];
                    const float* w = w4 + ((ci * 9) + dy * 3 + dx) * 24 + g * 6;
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[j] = fmaf(v, w[j], acc[j]);
                }
        }
        // skip1: 4x4 average of the gray tile (AvgPool2d(4,4)), then 1x1 conv 1->24 with bias
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += G[(4 * r + 6 + i) * GW + 4 * c + 6 + j];
        const float sk = s * 0.0625f;
        if (oy < H4 && ox < W4) {
            float* op = x1 + (((size_t)b * 24 + g * 6) * H4 + oy) * W4 + ox;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float v = fmaxf(acc[j], 0.f) + fmaf(sk, skw[g * 6 + j], skb[g * 6 + j]);
                op[(size_t)j * H4 * W4] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// block1 with its two 8-channel-input layers on the bf16 matrix cores (three-way split operands: fp32 arithmetic, k_conv_bx.hip).
// conv1 (1 -> 4) and conv2 (4 -> 8, s2) stay on the vector ALUs (K = 9 / 36 is no K for a matrix instruction); conv3 (8 -> 8) and conv4
// (8 -> 24, s2) have K = 72 = 9 taps x 8 channels: one tap per lane half = 5 K steps of 16 (the tenth tap is zero), six MFMAs each.
// Even with 8 of 32 output channels in use the matrix core does conv3 in a third of the VALU time, and the vector ALUs of the CU's other
// workgroup run conv1 / conv2 underneath.
//   * conv2's epilogue writes its 8 channels per pixel as three bf16 rows: C2s[19x35 pixels][split][8 ch] (48 B per pixel: the lanes of
//     a ds_read_b128 group stay on distinct banks); conv3's epilogue does the same for C3s[17x33] from its accumulators (4 channels
//     per lane half);
//   * conv3: 561 output pixels = 18 blocks of 32 consecutive pixels over the 8 waves; conv4: 4 blocks (2 output rows x 16 columns) on
//     waves 0-3; A = weights (cout in the lane, zero above 8 / 24), B = pixels, D: lane (pixel, half) holds couts (r&3) + 8(r>>2) + 4 half;
//   * persistent workgroups (two per CU): conv3's split weights stay in registers (60), conv4's are copied per tile into the part of
//     the dead c1 tile that C3s leaves free; C2s overlays the gray tile + the old c2 area (the skip average is taken first).
// ------------------------------------------------------------------------------------------------------------------------------
namespace b1x {
using namespace b1;
constexpr int PIXB = 48;                                              // bytes per pixel: 3 splits x 8 channels bf16
constexpr int C1B_OFF = 0, C1B_SZ = 4 * C1H * C1W * 4;                // 44304: c1 fp32 planes; later C3s + conv4 weights
constexpr int GB_OFF = C1B_OFF + C1B_SZ, GB_SZ = (GH * GW * 4 + 15) / 16 * 16;      // 11984: gray tile
constexpr int C2S_OFF = GB_OFF, C2S_SZ = C2H * C2W * PIXB;            // 31920: over the gray tile and beyond
constexpr int C3S_OFF = C1B_OFF, C3S_SZ = C3H * C3W * PIXB;           // 26928
constexpr int NF3 = 9, NF4 = 18;                                       // weight fragments (1 KiB each): 3 K steps x 3 splits (x 2 cout blocks for conv4)
constexpr int W4_OFF = (C3S_OFF + C3S_SZ + 15) / 16 * 16, W4_FIT = (C1B_OFF + C1B_SZ - W4_OFF) / 1024;      // 16 of conv4's 18 fragments fit behind C3s
constexpr int SK_OFF = C2S_OFF + (C2S_SZ > GB_SZ ? C2S_SZ : GB_SZ), SK_SZ = (128 + 8 + 24 + 48) * 4;      // skip averages, conv3 / conv4 biases, skip weights + biases
constexpr int W4B_OFF = (SK_OFF + SK_SZ + 15) / 16 * 16;              // the rest of conv4's fragments
constexpr int LDS_BYTES = W4B_OFF + (NF4 - W4_FIT) * 1024;           // 79.1 KB
static_assert(W4_FIT >= 1 && W4_FIT <= NF4 && LDS_BYTES <= 80 * 1024, "two workgroups per CU");
}  // namespace b1x

typedef float b1_f32x16 __attribute__((ext_vector_type(16)));
typedef float b1_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 b1_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b1_bf16x2 __attribute__((ext_vector_type(2)));
typedef float b1_f32x2 __attribute__((ext_vector_type(2)));

__device__ inline unsigned b1_pk_bf16(float a, float b) {
    const b1_f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b1_bf16x2));
}
__device__ inline void b1_split3(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = b1_pk_bf16(a, b);
    const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
    m = b1_pk_bf16(ra, rb);
    const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
    l = b1_pk_bf16(sa, sb);
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void block1_bx_kernel(const float* __restrict__ gray, const float* __restrict__ coef, float* __restrict__ x1, int B, int H, int W,
                                                        int tiles_x, int tiles_y,
                                                        const float* __restrict__ w1, const float* __restrict__ bb1,
                                                        const float* __restrict__ w2, const float* __restrict__ bb2,
                                                        const uint4* __restrict__ wq /* conv3: [3 steps][3 splits][64 lanes]; conv4: [2 cout blocks][3][3][64] */,
                                                        const float* __restrict__ bb3, const float* __restrict__ bb4,
                                                        const float* __restrict__ skw, const float* __restrict__ skb, long long* trace) {
    using namespace b1x;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b1[];
    float* G = reinterpret_cast<float*>(smem_b1 + GB_OFF);
    float* C1 = reinterpret_cast<float*>(smem_b1 + C1B_OFF);
    unsigned char* C2s = smem_b1 + C2S_OFF;
    unsigned char* C3s = smem_b1 + C3S_OFF;
    unsigned char* W4l = smem_b1 + W4_OFF;
    unsigned char* W4b = smem_b1 + W4B_OFF;
    float* SK = reinterpret_cast<float*>(smem_b1 + SK_OFF);
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H2 = H >> 1, W2 = W >> 1, H4 = H >> 2, W4 = W >> 2;
    const int tiles = tiles_x * tiles_y, total = tiles * B;

    if (tid < 8) SK[128 + tid] = bb3[tid];               // biases of the MFMA layers and the skip path's 1x1 conv: LDS, not registers held
    if (tid >= 64 && tid < 88) SK[136 + tid - 64] = bb4[tid - 64];      // across the whole kernel, and no global load behind a store in
    if (tid >= 128 && tid < 152) { SK[160 + tid - 128] = skw[tid - 128]; SK[184 + tid - 128] = skb[tid - 128]; }      // the epilogue
    const int ln = lane & 15, kg = lane >> 4;            // MFMA lane roles: pixel / cout index, tap group
    long long* tr = trace && tid == 0 ? trace + (size_t)blockIdx.x * 16 : nullptr;      // debug: stage stamps of the second tile
    int tix = 0;
#define B1_STAMP(k) { if (tr && tix == 1) tr[k] = __builtin_amdgcn_s_memtime(); }
    for (int vid = blockIdx.x; vid < total; vid += gridDim.x, ++tix) {
        int b, item;
        xcd_group_map(vid, tiles, B, b, item);
        const int Y4 = (item / tiles_x) * OH, X4 = (item % tiles_x) * OW;
        const float* gb = gray + (size_t)b * H * W;
        B1_STAMP(0)

        // ---- stage 0: gray tile, instance-normalised on the way in (zero padding stays zero) -------
        const float alpha = coef[2 * b], beta = coef[2 * b + 1];
        {
            constexpr int NL = (G_SZ + 511) / 512;
            float raw[NL];
            bool in[NL];
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const int e = tid + k * 512;
                const int r = e / GW, c = e - r * GW;
                const int gy = 4 * Y4 - 6 + r, gx = 4 * X4 - 6 + c;
                in[k] = e < G_SZ && gy >= 0 && gy < H && gx >= 0 && gx < W;
                raw[k] = in[k] ? gb[(size_t)gy * W + gx] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const int e = tid + k * 512;
                if (e < G_SZ) G[e] = in[k] ? fmaf(raw[k], alpha, beta) : 0.f;
            }
        }
        __syncthreads();
        B1_STAMP(1)

        // ---- stage 1: conv1 1->4, s1 (vector ALUs) ; the skip path's 4x4 averages while the gray tile is still there ----------
        for (int e = tid; e < C1H * C1W; e += 512) {
            const int r = e / C1W, c = e - r * C1W;
            const int gy = 4 * Y4 - 5 + r, gx = 4 * X4 - 5 + c;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
#pragma unroll
                for (int co = 0; co < 4; ++co) acc[co] = bb1[co];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float v = G[(r + dy) * GW + c + dx];
                        const float* w = w1 + (dy * 3 + dx) * 4;
#pragma unroll
                        for (int co = 0; co < 4; ++co) acc[co] = fmaf(v, w[co], acc[co]);
                    }
#pragma unroll
                for (int co = 0; co < 4; ++co) acc[co] = fmaxf(acc[co], 0.f);
            }
#pragma unroll
            for (int co = 0; co < 4; ++co) C1[co * (C1H * C1W) + e] = acc[co];
        }
        if (tid < 128) {
            const int r = tid >> 4, c = tid & 15;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) s += G[(4 * r + 6 + i) * GW + 4 * c + 6 + j];
            SK[tid] = s * 0.0625f;
        }
        __syncthreads();
        B1_STAMP(2)

        // conv3's split weights in operand order of v_mfma_f32_16x16x32_bf16: lane (cout = lane & 15, tap group kg = lane >> 4) holds tap
        // 4 s + kg of K step s (taps 9..11 are zero), channels 0..7: 3 steps x 3 splits = 9 fragments = 36 registers.  Loaded per tile
        // (L2-resident, in flight under conv2): held across the whole loop they cost scratch spills in every stage
        b1_bf16x8 w3[3][3];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int q = 0; q < 3; ++q) w3[s][q] = __builtin_bit_cast(b1_bf16x8, wq[(s * 3 + q) * 64 + lane]);
        uint4 w4r[3];              // conv4's split weights on their way to LDS (18 fragments = 1152 pieces of 16 B per workgroup)
#pragma unroll
        for (int k = 0; k < 3; ++k) w4r[k] = wq[NF3 * 64 + min(tid + 512 * k, NF4 * 64 - 1)];
        // ---- stage 2: conv2 4->8, s2 (vector ALUs); the 8 channels of a pixel leave as three bf16 rows -------------------------
        for (int e = tid; e < C2H * C2W; e += 512) {
            const int r = e / C2W, c = e - r * C2W;
            const int gy = 2 * Y4 - 2 + r, gx = 2 * X4 - 2 + c;
            float acc[8];
#pragma unroll
            for (int co = 0; co < 8; ++co) acc[co] = 0.f;
            if (gy >= 0 && gy < H2 && gx >= 0 && gx < W2) {
#pragma unroll
                for (int co = 0; co < 8; ++co) acc[co] = bb2[co];
#pragma unroll 1
                for (int ci = 0; ci < 4; ++ci) {
                    const float* src = C1 + ci * (C1H * C1W) + (2 * r) * C1W + 2 * c;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const float v = src[dy * C1W + dx];
                            const float* w = w2 + ((ci * 9) + dy * 3 + dx) * 8;
#pragma unroll
                            for (int co = 0; co < 8; ++co) acc[co] = fmaf(v, w[co], acc[co]);
                        }
                }
#pragma unroll
                for (int co = 0; co < 8; ++co) acc[co] = fmaxf(acc[co], 0.f);
            }
            uint4 h, m, l;
            b1_split3(acc[0], acc[1], h.x, m.x, l.x);
            b1_split3(acc[2], acc[3], h.y, m.y, l.y);
            b1_split3(acc[4], acc[5], h.z, m.z, l.z);
            b1_split3(acc[6], acc[7], h.w, m.w, l.w);
            unsigned char* p = C2s + e * PIXB;
            *reinterpret_cast<uint4*>(p) = h;
            *reinterpret_cast<uint4*>(p + 16) = m;
            *reinterpret_cast<uint4*>(p + 32) = l;
        }
        __syncthreads();           // c1 is dead from here: C3s and conv4's weights take its place
        B1_STAMP(3)

        // conv4's weights -> LDS (read in stage 4): loaded before conv2, stored now that c1 is dead
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int j = tid + 512 * k;          // fragment j >> 6: the first W4_FIT live behind C3s, the rest behind the skip table
            if (j < NF4 * 64) *reinterpret_cast<uint4*>((j < W4_FIT * 64 ? W4l + j * 16 : W4b + (j - W4_FIT * 64) * 16)) = w4r[k];
        }

        // ---- stage 3: conv3 8->8, s1 on the matrix cores: blocks of 16 consecutive pixels of the 17x33 tile, 18 MFMAs each, two blocks
        // at a time per wave (the 18 MFMAs of a block are one dependent accumulator chain: two chains keep the pipe fed) ----------------
        {
            constexpr int NBLK = (C3H * C3W + 15) / 16;      // 36
            auto conv3_blocks = [&](auto NBC, int blk0) __attribute__((always_inline)) {
                constexpr int NB = decltype(NBC)::value;
                int e[NB], r[NB], c[NB];
                const unsigned char* xb[NB];
                b1_f32x4 acc[NB];
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    e[k] = min((blk0 + 8 * k) * 16 + ln, C3H * C3W - 1);
                    r[k] = e[k] / C3W; c[k] = e[k] - r[k] * C3W;
                    xb[k] = C2s + (r[k] * C2W + c[k]) * PIXB;          // tap (dy, dx): + (dy * C2W + dx) * PIXB
                    acc[k] = b1_f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const int t = min(4 * s + kg, 8);                  // taps 9..11 re-read the ninth (zero weights)
                    const int toff = ((t / 3) * C2W + t % 3) * PIXB;
                    b1_bf16x8 x[NB][3];
#pragma unroll
                    for (int k = 0; k < NB; ++k)
#pragma unroll
                        for (int q = 0; q < 3; ++q) x[k][q] = *reinterpret_cast<const b1_bf16x8*>(xb[k] + toff + q * 16);
                    __builtin_amdgcn_sched_barrier(0);
                    // products (weight split, input split), small terms first; the blocks alternate
#define B1_MM(WQ, XQ) { _Pragma("unroll") for (int k = 0; k < NB; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3[s][WQ], x[k][XQ], acc[k], 0, 0, 0); }
                    B1_MM(2, 0) B1_MM(0, 2) B1_MM(1, 1) B1_MM(1, 0) B1_MM(0, 1) B1_MM(0, 0)
#undef B1_MM
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_nop 7\n\ts_nop 7");      // idle slots: whatever VALU code follows must not land in operand registers of the last MFMAs
                    __builtin_amdgcn_sched_barrier(0);
                }
                // D: lane (pixel ln, group kg) holds couts 4 kg .. + 3: groups 0, 1 are the 8 real channels.  Bias, ReLU, zero outside the
                // map, split, 8 bytes per split row
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const int gy = 2 * Y4 - 1 + r[k], gx = 2 * X4 - 1 + c[k];
                    const bool inmap = gy >= 0 && gy < H2 && gx >= 0 && gx < W2;
                    float y[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[j] = inmap ? fmaxf(acc[k][j] + SK[128 + ((4 * kg + j) & 7)], 0.f) : 0.f;
                    uint2 h, m, l;
                    b1_split3(y[0], y[1], h.x, m.x, l.x);
                    b1_split3(y[2], y[3], h.y, m.y, l.y);
                    if (kg < 2 && (blk0 + 8 * k) * 16 + ln < C3H * C3W) {
                        unsigned char* p = C3s + e[k] * PIXB + 8 * kg;
                        *reinterpret_cast<uint2*>(p) = h;
                        *reinterpret_cast<uint2*>(p + 16) = m;
                        *reinterpret_cast<uint2*>(p + 32) = l;
                    }
                }
            };
            int blk = wave;
            for (; blk + 8 < NBLK; blk += 16) conv3_blocks(std::integral_constant<int, 2>{}, blk);
            if (blk < NBLK) conv3_blocks(std::integral_constant<int, 1>{}, blk);
        }
        __syncthreads();
        B1_STAMP(4)

        // ---- stage 4: conv4 8->24, s2 on the matrix cores: wave w = output row w, 16 columns, two blocks of 16 couts + skip1 + add --
        {
            const int orow = wave, ocol = ln;
            const unsigned char* xb = C3s + ((2 * orow) * C3W + 2 * ocol) * PIXB;
            b1_f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            // (36 MFMAs per wave: no operand double-buffering here -- the sixteen waves of the CU cover each other's LDS latency, and the
            // registers stay under the 128 that four waves per SIMD allow)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int t = min(4 * s + kg, 8);
                const unsigned char* p = xb + ((t / 3) * C3W + t % 3) * PIXB;
                b1_bf16x8 x[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) x[q] = *reinterpret_cast<const b1_bf16x8*>(p + q * 16);
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    b1_bf16x8 w[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const int f = (cb * 3 + s) * 3 + q;            // compile-time after unrolling
                        w[q] = *reinterpret_cast<const b1_bf16x8*>((f < W4_FIT ? W4l + f * 1024 : W4b + (f - W4_FIT) * 1024) + lane * 16);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], x[0], acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[2], acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[1], acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[0], acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[1], acc[cb], 0, 0, 0);
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[0], acc[cb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            asm volatile("s_nop 7\n\ts_nop 7");
            __builtin_amdgcn_sched_barrier(0);
            // D: lane (pixel ln, group kg) holds couts 16 cb + 4 kg + j
            const int oy = Y4 + orow, ox = X4 + ocol;
            const float sk = SK[orow * 16 + ocol];
            if (oy < H4 && ox < W4) {
                float* op = x1 + ((size_t)b * 24 * H4 + oy) * W4 + ox;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int co = 16 * cb + 4 * kg + j;
                        if (co < 24) op[(size_t)co * H4 * W4] = fmaxf(acc[cb][j] + SK[136 + co], 0.f) + fmaf(sk, SK[160 + co], SK[184 + co]);
                    }
            }
        }
        B1_STAMP(5)
        __syncthreads();           // the tile's LDS is free for the next one
        B1_STAMP(6)
    }
#undef B1_STAMP
}

// A variant with ONLY conv4 on the matrix cores (same tiles and LDS budget as the kernel below, one tile per workgroup, nothing held in
// registers across stages) was no better: conv4 6.2 k -> 7.2 k cycles as written (its 36 MFMAs per wave are two dependent accumulator chains:
// ~150 cycles per v_mfma_f32_16x16x32_bf16 with the operand reads in between; four chains would bring it to ~3 k) while conv3's epilogue pays
// 2.6 k for splitting its outputs into bf16 rows -- a wash at best.  Removed.
// block1_bx_kernel (opt-in): per tile 5.3 k cycles gray + 7.4 k conv1 + 7.5 k conv2 (+ split) + 9-10 k conv3 + 4-10 k conv4 = 34-41 k against
// 32.9 k for the kernel below (325-371 us against 294).  conv3 / conv4 need 90 + 36 MFMAs per wave (~2 k cycles of the pipe) but at 128 VGPRs
// and ~100 SGPRs (two workgroups of 8 waves per CU, the VALU stages' scalar weight streams, a persistent loop, 14 pointer arguments) hipcc
// spills both register files, and every scratch reload parks a vmcnt(0) in the MFMA stages.  What it would take: conv3's weights in LDS
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
long long* g_block1_trace = nullptr;      // debug (xfh_debug_trace): stage stamps of block1_bx_kernel

void launch_block1_fused(const NetWeights& nw, const float* gray, const float* coef, int B, int H, int W, float* x1, hipStream_t st) {
    static int use_bx = -1;          // XFH_BLOCK1=bx: conv3 / conv4 on split-bf16 MFMAs (block1_bx_kernel; correct, slower: see the note below)
    if (use_bx < 0) { const char* e = getenv("XFH_BLOCK1"); use_bx = e && !strcmp(e, "bx") ? 1 : 0; }
    if (use_bx && nw.block1_bx) {
        static unsigned attr = 0;
        set_max_dynamic_lds(reinterpret_cast<const void*>(block1_bx_kernel), b1x::LDS_BYTES, attr);
        const int H4 = H / 4, W4 = W / 4;
        const int tx = ceil_div(W4, b1::OW), ty = ceil_div(H4, b1::OH);
        int grid = 2 * num_cus();
        if (grid > tx * ty * B) grid = tx * ty * B;
        block1_bx_kernel<<<grid, 512, b1x::LDS_BYTES, st>>>(gray, coef, x1, B, H, W, tx, ty, nw.conv[L_BLOCK1_0].w_kc, nw.conv[L_BLOCK1_0].bias,
                                                            nw.conv[L_BLOCK1_1].w_kc, nw.conv[L_BLOCK1_1].bias, reinterpret_cast<const uint4*>(nw.block1_bx),
                                                            nw.conv[L_BLOCK1_2].bias, nw.conv[L_BLOCK1_3].bias, nw.conv[L_SKIP1].w_oihw, nw.conv[L_SKIP1].bias, g_block1_trace);
        return;
    }
    const ConvW& c0 = nw.conv[L_BLOCK1_0];
    const ConvW& c1 = nw.conv[L_BLOCK1_1];
    const ConvW& c2 = nw.conv[L_BLOCK1_2];
    const ConvW& c3 = nw.conv[L_BLOCK1_3];
    const ConvW& sk = nw.conv[L_SKIP1];
    const int H4 = H / 4, W4 = W / 4;
    const int tx = ceil_div(W4, b1::OW), ty = ceil_div(H4, b1::OH);
    static unsigned attr = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(block1_fused_kernel), b1::LDS_FLOATS * 4, attr);
    block1_fused_kernel<<<xcd_grid_size(tx * ty, B), 512, b1::LDS_FLOATS * 4, st>>>(
        gray, coef, x1, B, H, W, tx, ty, c0.w_kc, c0.bias, c1.w_kc, c1.bias, c2.w_kc, c2.bias, c3.w_kc, c3.bias, sk.w_oihw, sk.bias);
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
