// 3x3 / 1x1 convolution as an implicit GEMM on the f32 matrix cores (v_mfma_f32_32x32x2_f32),
// optionally with the FOLLOWING 1x1 convolution fused in.
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
// = 64 VGPRs).  K is walked in chunks of CK input channels.  A chunk's weights
// [CK*k*k][COUT_PAD] and its input tile with halo [CK][plane] are copied global -> LDS by the
// DMA path (global_load_lds: no VGPR staging, no ds_write) into one of TWO LDS buffers: the
// copy of chunk c+1 flies while chunk c feeds the MFMAs, one barrier per chunk.  Zero padding
// comes from pointing out-of-image lanes at a page of zeros.
//
// Fused trailing 1x1 (block3.1+3.2, block5.2+5.3, block_fusion.1+.2): the K pairing of an MFMA
// is free as long as A and B agree.  After bias+ReLU the accumulator register r of block m
// holds, for this lane's pixel, channel m*32+(r&3)+8*(r>>2)+4*(l>>5) -- so pairing channels
// (c, c+4) makes those registers THE B operand of the 1x1 GEMM: no LDS round trip, no
// shuffles; only the 1x1 weights [K][COUT2] sit in LDS.
#include "kernels.hpp"
#include <cstdlib>

namespace xfh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct ConvGeom {
    int Hin, Win, Hout, Wout;
    int TW, TH, tiles_x, tiles, B;
    int IWt, plane, PS;     // input tile width, elements, plane stride in LDS (multiple of 64)
};

struct ConvArgs {
    const float* in;
    const float* wk;        // [CIN*KK][COUT_PAD]
    const float* bias;      // [COUT_PAD]
    float* out;
    const float* zeros;     // >= 64 B of zeros (padding source for the DMA)
    const float* wk2;       // fused 1x1: [COUT][COUT2_PAD]
    const float* bias2;
    int relu, relu2;
    long long* trace;       // debug: per-workgroup s_memtime stamps (NULL in production)
    ConvGeom g;
    int cold;
};

template <int CIN, int COUT, int KS, int STRIDE, int CK, int NSEG, bool NHWC, int COUT2, int NBO = 0>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
    kernel_entry_hooks(a.cold);      // debug: code-position shift / cold instruction cache (common.hpp)
    constexpr int COUT_PAD = (COUT + 31) / 32 * 32;
    constexpr int MB = COUT_PAD / 32;
    constexpr int NB = NBO > 0 ? NBO : (MB == 1 ? 4 : (MB == 2 ? 2 : 1));
    constexpr int KK = KS * KS;
    constexpr int PAD = KS / 2;
    constexpr int NCH = CIN / CK;
    static_assert(CIN % CK == 0 && CK % 2 == 0, "CIN must be a multiple of the (even) channel chunk");
    constexpr int WCH = CK * KK * COUT_PAD;          // floats per weight chunk
    static_assert(WCH % 4 == 0, "weight chunk must be whole 16-byte DMA elements");
    constexpr int COUT2_PAD = (COUT2 + 31) / 32 * 32;
    constexpr int MB2 = COUT2_PAD / 32;
    static_assert(COUT2 == 0 || COUT == COUT_PAD, "fused 1x1 needs a 32-multiple channel count");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    const ConvGeom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b, tile;                      // all tiles of an image on one XCD (halo rows + weights hit its L2)
    if (!xcd_group_map(blockIdx.x, g.tiles, g.B, b, tile)) return;
    const int tyi = tile / g.tiles_x, txi = tile % g.tiles_x;
    const int oy0 = tyi * g.TH, ox0 = txi * g.TW;
    const int npix = g.TH * g.TW;
    const int PS = g.PS;
    const int SB = WCH + CK * PS;                    // floats per staging buffer
    const size_t HWin = (size_t)g.Hin * g.Win;
    const float* inb = a.in + (size_t)b * CIN * HWin;

    long long* tr = a.trace ? a.trace + (size_t)blockIdx.x * 24 : nullptr;
    if (tr && tid == 0) { tr[0] = __builtin_amdgcn_s_memtime(); tr[19] = __builtin_amdgcn_s_memrealtime(); }
    // fused 1x1 weights: one DMA at kernel start into their own LDS region
    if (COUT2 > 0) {
        float* W2l = smem + 2 * SB;
        for (int j = wave; j < COUT * COUT2_PAD / 256; j += 4)
            __builtin_amdgcn_global_load_lds((gptr_t)(a.wk2 + j * 256 + lane * 4), (lptr_t)(W2l + j * 256), 16, 0, 0);
    }

    // this wave copies plane segments seg = wave + 4*s (64 elements each) of every channel.
    // Out-of-image (and tail) lanes read the zero page with a zero channel stride: branch-free.
    const float* xsrc[NSEG];
    unsigned xmask[NSEG];
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        const int e = (wave + 4 * s) * 64 + lane;
        int go = -1;
        if (e < g.plane) {
            const int r = e / g.IWt, c = e - r * g.IWt;
            const int gy = oy0 * STRIDE - PAD + r, gx = ox0 * STRIDE - PAD + c;
            if (gy >= 0 && gy < g.Hin && gx >= 0 && gx < g.Win) go = gy * g.Win + gx;
        }
        xsrc[s] = go >= 0 ? inb + go : a.zeros + lane;
        xmask[s] = go >= 0 ? 0xffffffffu : 0u;
    }
    const unsigned HWu = (unsigned)HWin;
    auto issue = [&](int ch, int bsel) {
        float* Wd = smem + bsel * SB;
        float* Xd = Wd + WCH;
        const float* wsrc = a.wk + (size_t)ch * WCH;
#pragma unroll
        for (int jj = 0; jj < (WCH / 256 + 3) / 4; ++jj) {
            const int j = wave + 4 * jj;               // whole 1 KiB pieces: wave-uniform condition
            if (j < WCH / 256)
                __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + j * 256 + lane * 4), (lptr_t)(Wd + j * 256), 16, 0, 0);
        }
        if (WCH % 256 != 0 && wave == (WCH / 256) % 4 && lane * 4 < WCH % 256)      // partial last piece
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (WCH / 256) * 256 + lane * 4), (lptr_t)(Wd + (WCH / 256) * 256), 16, 0, 0);
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
            const int seg = wave + 4 * s;
            if (seg * 64 < PS) {
#pragma unroll
                for (int c = 0; c < CK; ++c) {
                    const unsigned off = ((unsigned)(ch * CK + c) * HWu) & xmask[s];
                    __builtin_amdgcn_global_load_lds((gptr_t)(xsrc[s] + off), (lptr_t)(Xd + c * PS + seg * 64), 4, 0, 0);
                }
            }
        }
    };

    // LDS offset of this lane's pixel for each of its pixel blocks (k-half folded in)
    int pixoff[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        int t = (wave * NB + n) * 32 + l31;
        if (t >= npix) t = 0;
        const int ty = t / g.TW, tx = t - ty * g.TW;
        pixoff[n] = WCH + half * PS + ty * STRIDE * g.IWt + tx * STRIDE;
    }
    const int wbase = half * KK * COUT_PAD + l31;

    // accumulators start at the folded-BN shift / bias (loaded here, before any store, so the
    // loads batch; an epilogue load per store would serialise on vmcnt(0))
    f32x16 acc[MB][NB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // D rows are channels (NCHW / fused) or pixels (plain NHWC: columns are channels)
            const float bs = (NHWC && COUT2 == 0) ? a.bias[m * 32 + l31] : a.bias[m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[m][n][r] = bs;
        }

    issue(0, 0);
    if (tr && tid == 0) tr[1] = __builtin_amdgcn_s_memtime();
    for (int ch = 0; ch < NCH; ++ch) {
        lds_dma_barrier();   // chunk ch has landed (explicit vmcnt(0) + barrier); buffer (ch+1)&1 is free again
        if (tr && tid == 0 && ch < 17) tr[2 + ch] = __builtin_amdgcn_s_memtime();
        if (ch + 1 < NCH) issue(ch + 1, (ch + 1) & 1);
        const float* S = smem + (ch & 1) * SB;
        // software pipeline over the NS = (CK/2)*k*k MFMA steps of this chunk: the operands of
        // step st+1 are read from LDS into the other register set BEFORE the MFMAs of step st
        // issue, so a lone wave never exposes the LDS latency (hipcc left to itself emits
        // ds_read -> lgkmcnt(0) -> MFMAs per step on one register set).
        constexpr int NS = (CK / 2) * KK;
        float av[2][MB], bv[2][NB];
        auto ld = [&](int st, float (&ao)[MB], float (&bo)[NB]) {
            const int p = st / KK, tap = st - p * KK;
            const int toff = KS == 3 ? (tap / 3) * g.IWt + (tap % 3) : 0;
#pragma unroll
            for (int m = 0; m < MB; ++m) ao[m] = S[wbase + ((2 * p) * KK + tap) * COUT_PAD + m * 32];
#pragma unroll
            for (int n = 0; n < NB; ++n) bo[n] = S[pixoff[n] + 2 * p * PS + toff];
        };
        // Wanted issue order:  ld(0) | MFMA(st)#1, ld(st+1), MFMA(st)#2.. | ...  -- the reads of
        // the next step go right behind the FIRST MFMA of the current step: they issue while the
        // matrix pipe is busy (a wave blocked on its 2nd MFMA cannot issue anything else) and
        // have ~190 cycles to land before they are needed.
        ld(0, av[0], bv[0]);
        __builtin_amdgcn_sched_group_barrier(0x100, MB + NB, 0);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            if (st + 1 < NS) ld(st + 1, av[(st + 1) & 1], bv[(st + 1) & 1]);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    acc[m][n] = (NHWC && COUT2 == 0)
                                    ? __builtin_amdgcn_mfma_f32_32x32x2f32(bv[st & 1][n], av[st & 1][m], acc[m][n], 0, 0, 0)
                                    : __builtin_amdgcn_mfma_f32_32x32x2f32(av[st & 1][m], bv[st & 1][n], acc[m][n], 0, 0, 0);
            if (st + 1 < NS) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, MB + NB, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MB * NB - 1, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, MB * NB, 0);
            }
        }
    }

    if (tr && tid == 0) tr[20] = __builtin_amdgcn_s_memtime();
    if (COUT2 == 0) {
        // ---- epilogue: + folded-BN shift / bias, ReLU, store --------------------------------
        if (!NHWC) {
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const int t = (wave * NB + n) * 32 + l31;
                const int ty = t / g.TW, tx = t - ty * g.TW;
                const int oy = oy0 + ty, ox = ox0 + tx;
                const bool ok = t < npix && oy < g.Hout && ox < g.Wout;
                float* op = a.out + ((size_t)b * COUT * g.Hout + oy) * g.Wout + ox;
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (ok && co < COUT) {
                            float v = acc[m][n][r];
                            if (a.relu) v = fmaxf(v, 0.f);
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
                    float* op = a.out + (((size_t)b * g.Hout + oy) * g.Wout + ox) * COUT;
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        const int co = m * 32 + l31;
                        if (ok && co < COUT) {
                            float v = acc[m][n][r];
                            if (a.relu) v = fmaxf(v, 0.f);
                            op[co] = v;
                        }
                    }
                }
        }
        if (tr && tid == 0) { tr[21] = __builtin_amdgcn_s_memtime(); tr[23] = __builtin_amdgcn_s_memrealtime(); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); tr[22] = ((long long)xcc << 32) | hwid; }
        return;
    }

    // ---- fused 1x1: y = relu(acc + bias) stays in registers and IS the B operand ------------
    if (COUT2 > 0) {
        const float* W2l = smem + 2 * SB;
        f32x16 acc2[MB2 > 0 ? MB2 : 1][NB];
#pragma unroll
        for (int m2 = 0; m2 < MB2; ++m2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bs2 = NHWC ? a.bias2[m2 * 32 + l31] : a.bias2[m2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
                for (int n = 0; n < NB; ++n) acc2[m2][n][r] = bs2;
            }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kc = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;   // this lane's channel for step (m,r)
                float av2[MB2 > 0 ? MB2 : 1];
#pragma unroll
                for (int m2 = 0; m2 < MB2; ++m2) av2[m2] = W2l[kc * COUT2_PAD + m2 * 32 + l31];
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    float y = acc[m][n][r];
                    if (a.relu) y = fmaxf(y, 0.f);
#pragma unroll
                    for (int m2 = 0; m2 < MB2; ++m2)
                        acc2[m2][n] = NHWC ? __builtin_amdgcn_mfma_f32_32x32x2f32(y, av2[m2], acc2[m2][n], 0, 0, 0)
                                           : __builtin_amdgcn_mfma_f32_32x32x2f32(av2[m2], y, acc2[m2][n], 0, 0, 0);
                }
            }
        }
        if (!NHWC) {
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const int t = (wave * NB + n) * 32 + l31;
                const int ty = t / g.TW, tx = t - ty * g.TW;
                const int oy = oy0 + ty, ox = ox0 + tx;
                const bool ok = t < npix && oy < g.Hout && ox < g.Wout;
                float* op = a.out + ((size_t)b * COUT2 * g.Hout + oy) * g.Wout + ox;
#pragma unroll
                for (int m2 = 0; m2 < MB2; ++m2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = m2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (ok && co < COUT2) {
                            float v = acc2[m2][n][r];
                            if (a.relu2) v = fmaxf(v, 0.f);
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
                    float* op = a.out + (((size_t)b * g.Hout + oy) * g.Wout + ox) * COUT2;
#pragma unroll
                    for (int m2 = 0; m2 < MB2; ++m2) {
                        const int co = m2 * 32 + l31;
                        if (ok && co < COUT2) {
                            float v = acc2[m2][n][r];
                            if (a.relu2) v = fmaxf(v, 0.f);
                            op[co] = v;
                        }
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static bool choose_tile(int Hout, int Wout, int tile_pix, int S, int KS, int nseg, ConvGeom& g) {
    long best_tiles = -1, best_cost = 0;
    for (int tw = 1; tw <= Wout && tw <= tile_pix; ++tw) {
        int th = tile_pix / tw;
        if (th > Hout) th = Hout;
        if (th < 1) continue;
        const int iwt = (tw - 1) * S + KS, iht = (th - 1) * S + KS;
        const long plane = (long)iwt * iht;
        if (plane > (long)nseg * 256) continue;
        const long tiles = (long)ceil_div(Hout, th) * ceil_div(Wout, tw);
        const long cost = tiles * plane;
        if (best_tiles < 0 || tiles < best_tiles || (tiles == best_tiles && cost < best_cost)) {
            best_tiles = tiles; best_cost = cost;
            g.TW = tw; g.TH = th; g.IWt = iwt; g.plane = (int)plane;
        }
    }
    if (best_tiles < 0) return false;
    g.tiles_x = ceil_div(Wout, g.TW);
    g.PS = ceil_div(g.plane, 64) * 64;
    return true;
}

template <int CIN, int COUT, int KS, int STRIDE, int CK, int NSEG, int COUT2, int NBO = 0>
static int run(const ConvW& c, const ConvW* c2, const float* zeros, const float* in, int B, int Hin, int Win, float* out,
               bool nhwc, hipStream_t st, long long* trace) {
    constexpr int COUT_PAD = (COUT + 31) / 32 * 32;
    constexpr int MB = COUT_PAD / 32;
    constexpr int NB = NBO > 0 ? NBO : (MB == 1 ? 4 : (MB == 2 ? 2 : 1));
    constexpr int WCH = CK * KS * KS * COUT_PAD;
    constexpr int COUT2_PAD = (COUT2 + 31) / 32 * 32;
    ConvArgs a;
    a.cold = g_debug_cold;
    ConvGeom& g = a.g;
    g.Hin = Hin; g.Win = Win;
    g.Hout = (Hin + 2 * (KS / 2) - KS) / STRIDE + 1;
    g.Wout = (Win + 2 * (KS / 2) - KS) / STRIDE + 1;
    if (!choose_tile(g.Hout, g.Wout, 4 * NB * 32, STRIDE, KS, NSEG, g)) return -1;
    a.in = in; a.wk = c.w_kcp; a.bias = c.bias; a.out = out; a.zeros = zeros; a.relu = c.relu;
    a.trace = trace;
    a.wk2 = c2 ? c2->w_kcp : nullptr; a.bias2 = c2 ? c2->bias : nullptr; a.relu2 = c2 ? c2->relu : 0;
    const int tiles = g.tiles_x * ceil_div(g.Hout, g.TH);
    g.tiles = tiles; g.B = B;
    const size_t lds = ((size_t)2 * (WCH + CK * g.PS) + (size_t)COUT * COUT2_PAD * (COUT2 > 0)) * sizeof(float);
    if (lds > 160 * 1024) return -1;
    static AttrMask attr_a{0}, attr_b{0};     // per instantiation of run<>, per device
    set_max_dynamic_lds(reinterpret_cast<const void*>(conv_mfma_kernel<CIN, COUT, KS, STRIDE, CK, NSEG, true, COUT2, NBO>), 160 * 1024, attr_a);
    set_max_dynamic_lds(reinterpret_cast<const void*>(conv_mfma_kernel<CIN, COUT, KS, STRIDE, CK, NSEG, false, COUT2, NBO>), 160 * 1024, attr_b);
    if (nhwc) conv_mfma_kernel<CIN, COUT, KS, STRIDE, CK, NSEG, true, COUT2, NBO><<<xcd_grid_size(tiles, B), 256, lds, st>>>(a);
    else conv_mfma_kernel<CIN, COUT, KS, STRIDE, CK, NSEG, false, COUT2, NBO><<<xcd_grid_size(tiles, B), 256, lds, st>>>(a);
    return 0;
}

int launch_conv_mfma(const ConvW& c, const ConvW* fused1x1, const float* zeros, const float* in, int B, int Hin, int Win,
                     float* out, bool nhwc, hipStream_t st, long long* trace) {
    const int key = c.cin * 1000000 + c.cout * 1000 + c.ks * 10 + c.stride;
    if (fused1x1) {
        if (fused1x1->ks != 1 || fused1x1->cin != c.cout) return -1;
        if (key == 64 * 1000000 + 64 * 1000 + 31 && fused1x1->cout == 64) {
            return run<64, 64, 3, 1, 8, 2, 64>(c, fused1x1, zeros, in, B, Hin, Win, out, nhwc, st, trace);
        }
        if (key == 128 * 1000000 + 128 * 1000 + 31 && fused1x1->cout == 64)
            return run<128, 128, 3, 1, 4, 1, 64>(c, fused1x1, zeros, in, B, Hin, Win, out, nhwc, st, trace);
        return -1;
    }
    // small maps: 128-pixel tiles (one pixel block per wave) give 2x the workgroups, each half as
    // heavy -> less tail and more co-resident workgroups on the 30x40 / 15x20 maps
    const int Hout = (Hin + 2 * (c.ks / 2) - c.ks) / c.stride + 1, Wout = (Win + 2 * (c.ks / 2) - c.ks) / c.stride + 1;
    const bool small_map = (long)B * Hout * Wout <= 160L * 1024;
    switch (key) {
        case 24 * 1000000 + 24 * 1000 + 31:
            // 256-pixel tiles (2 pixel blocks per wave): A/B measured 1 % faster than 512-pixel tiles
            return run<24, 24, 3, 1, 4, 2, 0, 2>(c, nullptr, zeros, in, B, Hin, Win, out, nhwc, st, trace);
        case 24 * 1000000 + 64 * 1000 + 32:
            return run<24, 64, 3, 2, 4, 5, 0>(c, nullptr, zeros, in, B, Hin, Win, out, nhwc, st, trace);
        case 64 * 1000000 + 64 * 1000 + 31:
            if (small_map) return run<64, 64, 3, 1, 4, 1, 0, 1>(c, nullptr, zeros, in, B, Hin, Win, out, nhwc, st, trace);
            return run<64, 64, 3, 1, 8, 2, 0>(c, nullptr, zeros, in, B, Hin, Win, out, nhwc, st, trace);
        case 64 * 1000000 + 64 * 1000 + 32:
            if (small_map) return run<64, 64, 3, 2, 4, 3, 0, 1>(c, nullptr, zeros, in, B, Hin, Win, out, nhwc, st, trace);
            return run<64, 64, 3, 2, 4, 5, 0>(c, nullptr, zeros, in, B, Hin, Win, out, nhwc, st, trace);
        case 64 * 1000000 + 128 * 1000 + 32:  return run<64, 128, 3, 2, 4, 3, 0>(c, nullptr, zeros, in, B, Hin, Win, out, nhwc, st, trace);
        case 128 * 1000000 + 128 * 1000 + 31: return run<128, 128, 3, 1, 4, 1, 0>(c, nullptr, zeros, in, B, Hin, Win, out, nhwc, st, trace);
        case 64 * 1000000 + 64 * 1000 + 11:   return run<64, 64, 1, 1, 32, 1, 0>(c, nullptr, zeros, in, B, Hin, Win, out, nhwc, st, trace);
        case 128 * 1000000 + 64 * 1000 + 11:  return run<128, 64, 1, 1, 32, 1, 0>(c, nullptr, zeros, in, B, Hin, Win, out, nhwc, st, trace);
    }
    return -1;
}

double conv_flops(const ConvW& c, int B, int Hout, int Wout) {
    return 2.0 * B * Hout * Wout * (double)c.cout * c.cin * c.ks * c.ks;
}

}  // namespace xfh
