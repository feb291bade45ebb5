// The f32-MFMA heads with register input (head_f32r_kernel, k_heads.hip: the DEFAULT heads) and the chained layer they share with head_fused_kernel: bodies in a
// header of their own so that tests/emu/ can compile the SAME source for the host (XFH_HOST_EMU) and run it against a float64 reference without a GPU.
#pragma once
#ifndef XFH_HOST_EMU
#include "kernels.hpp"
#ifndef XFH_DYN_LDS
#define XFH_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) float name[]
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

struct HeadArgs {
    const float* src;        // KP: raw gray (B,H,W) ; REL: feats (B*hc*wc, 64)
    const float* coef;       // KP: per-image instance-norm {alpha, beta}
    const float* zeros;
    const float* w[4];       // [64][n_pad] per layer (BN folded)
    const float* bias[4];
    float* out;              // KP: heat (B,H,W) ; REL: reliability (B*hc*wc)
    float* logits;           // KP only, optional: (B*hc*wc, 65)
    float* inv;              // REL only, optional: 1 / max(||feats[cell,:]||, 1e-12)  (F.normalize(M1, dim=1), xfeat.py:70)
    int H, W, hc, wc, ncell, ntiles;
    int cold;
};

// one chained 64 -> 32*MBO layer: out = bias + W^T relu(in)   (in/out in D[feature][cell] layout)
// DUST (the key-point head's last layer, MBO = 2): output 64 -- the dustbin logit, the only real row of what would be a third block of 32 -- is taken as a dot
// product on the vector ALUs: its weight of the step's channel is one more LDS read (the same address for the 32 lanes of a half: a broadcast) and one fma per K step
// instead of an MFMA per K step; `dust` receives this lane's half of the sum (its 32 channels; the other half-wave holds the other 32)
template <int MBO, bool DUST = false>
__device__ inline void chain_layer(const float* __restrict__ Wl, int npad, const float* __restrict__ bias,
                                   const f32x16 (&in)[2], f32x16 (&out)[MBO], int l31, int half, float* dust = nullptr) {
#pragma unroll
    for (int m = 0; m < MBO; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[m][r] = bias[m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
    const float* wb = Wl + (4 * half) * npad + l31;     // this lane's channel offset (c or c+4)
    const float* wd = Wl + (4 * half) * npad + 64;      // DUST: column 64 of the same rows
    float av[2][MBO], dv[2] = {0.f, 0.f}, ds = 0.f;
    auto ld = [&](int st, float (&ao)[MBO], float& dvo) {
        const int m = st >> 4, r = st & 15;
        const int k0 = m * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int mo = 0; mo < MBO; ++mo) ao[mo] = wb[k0 * npad + mo * 32];
        if constexpr (DUST) dvo = wd[k0 * npad];
    };
    ld(0, av[0], dv[0]);
    __builtin_amdgcn_sched_group_barrier(0x100, MBO + DUST, 0);
#pragma unroll
    for (int st = 0; st < 32; ++st) {
        if (st + 1 < 32) ld(st + 1, av[(st + 1) & 1], dv[(st + 1) & 1]);
        const float y = fmaxf(in[st >> 4][st & 15], 0.f);
        if constexpr (DUST) ds = fmaf(y, dv[st & 1], ds);
#pragma unroll
        for (int mo = 0; mo < MBO; ++mo)
            out[mo] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st & 1][mo], y, out[mo], 0, 0, 0);
        if (st + 1 < 32) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, MBO + DUST, 0);
            if (MBO > 1) __builtin_amdgcn_sched_group_barrier(0x008, MBO - 1, 0);
        } else {
            __builtin_amdgcn_sched_group_barrier(0x008, MBO, 0);
        }
    }
    if constexpr (DUST) *dust = ds;
}

// ------------------------------------------------------------------------------------------------------------------------------
// The f32-MFMA heads without the activation tile (round 4; option heads_f32 = 2).  head_fused_kernel stages a tile's first-layer input in LDS and
// needs two barriers per tile for it: its eight waves run every phase together, and the softmax / store epilogue of all of them meets idle matrix
// cores (157 + 61 us per 64-frame step against ~105 us of f32 MFMA work).  Here the first layer's B operand comes straight from registers: with the K
// order  step p = 4 dy + i, lane half h  <->  channel 8 dy + 4 h + i  a lane's 32 channels are eight float4 (half a pixel row of the 8x8 cell for the
// unfold; half of every 8-channel group of the channels-last feature row), loaded one tile ahead; the weights' LDS address follows the same order.
// After the weights have landed there is no barrier: the waves drift apart and cover each other's epilogues, as in head_bx_kernel -- on the f32
// instruction (v_mfma_f32_32x32x2_f32, one VGPR per operand), which the cold-instruction-cache torture of tools/head_soak.py does not trip.
// ------------------------------------------------------------------------------------------------------------------------------
template <bool KP, bool DUST = true>      // DUST = false: the round-4 form (the dustbin logit as the only real row of a third cout block), kept for A/B (heads_f32 = 3)
__device__ __forceinline__ void head_f32r_body(const HeadArgs& a) {
    constexpr int NL = KP ? 4 : 3;
    XFH_DYN_LDS(smem_r);
    float* Wl = smem_r;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = a.hc * a.wc;
    {
        int off = 0;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int n = (KP && l == 3) ? 64 * 96 : ((!KP && l == 2) ? 64 : 64 * 64);
            if (n >= 256) {
                for (int j = wave; j < n / 256; j += 8)
                    __builtin_amdgcn_global_load_lds((xfh_gptr_t)(a.w[l] + j * 256 + lane * 4), (xfh_lptr_t)(Wl + off + j * 256), 16, 0, 0);
            } else if (wave == 0) {
                __builtin_amdgcn_global_load_lds((xfh_gptr_t)(a.w[l] + lane), (xfh_lptr_t)(Wl + off), 4, 0, 0);
            }
            off += n;
        }
    }
    float4 xin[8];
    float nalpha = 1.f, nbeta = 0.f;                                   // (of the tile xin belongs to)
    auto issue_x = [&](int tile) __attribute__((always_inline)) {
        const int g = min(tile * HD_CELLS + wave * 32 + l31, a.ncell - 1);      // cells past the end: copies of the last one, never stored
        const float* p;
        size_t step;
        if (KP) {
            const int b = g / hw, rem = g - b * hw;
            const int ci = rem / a.wc, cj = rem - ci * a.wc;
            p = a.src + (size_t)b * a.H * a.W + (size_t)(8 * ci) * a.W + 8 * cj + 4 * half;      // pixel row dy, columns 4 h .. 4 h + 3
            step = (size_t)a.W;
            nalpha = a.coef[2 * b]; nbeta = a.coef[2 * b + 1];
        } else {
            p = a.src + (size_t)g * 64 + 4 * half;
            step = 8;
        }
#pragma unroll
        for (int dy = 0; dy < 8; ++dy) xin[dy] = *reinterpret_cast<const float4*>(p + dy * step);
    };
    int tile = blockIdx.x;
    if (tile < a.ntiles) issue_x(tile);
    lds_dma_barrier();                                                // the weights have landed; no barrier from here on
    for (; tile < a.ntiles; tile += gridDim.x) {
        const int gcell = tile * HD_CELLS + wave * 32 + l31;          // this lane's cell
        f32x16 accA[2], accB[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) accA[m][r] = a.bias[0][m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
        {
            const float* wb = Wl + (4 * half) * 64 + l31;               // step p -> channel 8 (p >> 2) + 4 half + (p & 3)
            const float al = nalpha, be = nbeta;
            float av[2][2];
            auto ld = [&](int p, float (&ao)[2]) {
                const int k = 8 * (p >> 2) + (p & 3);
                ao[0] = wb[k * 64];
                ao[1] = wb[k * 64 + 32];
            };
            ld(0, av[0]);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            float nrm2 = 0.f;          // REL: this lane walks 32 of its cell's 64 channels anyway -> squared norm for free
#pragma unroll
            for (int p = 0; p < 32; ++p) {
                if (p + 1 < 32) ld(p + 1, av[(p + 1) & 1]);
                const float4 q = xin[p >> 2];
                const float raw = (p & 3) == 0 ? q.x : (p & 3) == 1 ? q.y : (p & 3) == 2 ? q.z : q.w;
                const float xv = KP ? fmaf(raw, al, be) : raw;
                if (!KP) nrm2 = fmaf(xv, xv, nrm2);
                accA[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p & 1][0], xv, accA[0], 0, 0, 0);
                accA[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p & 1][1], xv, accA[1], 0, 0, 0);
                if (p + 1 < 32) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
            }
            if (!KP && a.inv) {
                nrm2 += xhalf(nrm2);                       // the other 32 channels sit in the other half-wave
                if (half == 0 && gcell < a.ncell) a.inv[gcell] = 1.f / fmaxf(sqrtf(nrm2), 1e-12f);
            }
        }
        if (tile + (int)gridDim.x < a.ntiles) issue_x(tile + gridDim.x);      // the next tile's input flies during the chained layers
        if (KP) {
            chain_layer<2>(Wl + 64 * 64, 64, a.bias[1], accA, accB, l31, half);
            chain_layer<2>(Wl + 2 * 64 * 64, 64, a.bias[2], accB, accA, l31, half);
            f32x16 lg[DUST ? 2 : 3];
            float lgd;
            if constexpr (DUST) {
                float dust;
                chain_layer<2, true>(Wl + 3 * 64 * 64, 96, a.bias[3], accA, lg, l31, half, &dust);      // (the 64 real outputs on the matrix cores, the dustbin logit as a dot product)
                lgd = dust + xhalf(dust) + a.bias[3][64];
            } else {
                chain_layer<3>(Wl + 3 * 64 * 64, 96, a.bias[3], accA, lg, l31, half);
                lgd = lg[DUST ? 0 : 2][0];          // (half 0 only: the one place that uses it)
            }
            // lane (l31,half) holds logits c = 32m + (r&3) + 8(r>>2) + 4*half of its cell; the dustbin logit (c == 64) is lgd, in every lane
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
        } else {
            chain_layer<2>(Wl + 64 * 64, 64, a.bias[1], accA, accB, l31, half);
            const float* w3 = Wl + 2 * 64 * 64;
            float sdot = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sdot = fmaf(fmaxf(accB[m][r], 0.f), w3[m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half], sdot);
            sdot += __shfl_xor(sdot, 32, 64);
            if (half == 0 && gcell < a.ncell) a.out[gcell] = 1.f / (1.f + expf(-(sdot + a.bias[2][0])));
        }
    }
}



}  // namespace xfh
