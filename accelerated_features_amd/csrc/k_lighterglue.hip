// LighterGlue (kornia LightGlue under modules/lighterglue.py:12-27: d = 96, one head, 6 layers) -- device kernels.
// Call sites in the reference: modules/lighterglue.py:50-57, modules/xfeat.py:131-162.  Arithmetic: the published
// LightGlue v0.1 algorithm (kornia 0.7.2 is not on disk; DESIGN.md says what the parity claim is pinned to).
//
// One pair per call (the reference supports B = 1 only).  Everything keeps a fixed capacity N and a device-side
// live count (width pruning shrinks the sets after every layer): no host read-back until the final match list.
//
//   lg_encode_kernel      key-point normalisation + learnable Fourier encoding -> cos / sin tables (N, 96)
//   lg_linear_kernel      y = x W^T + b, register-direct on v_mfma_f32_32x32x2_f32 (no LDS: one wave = 32 rows x 32
//                         output features, operands straight from L2 in operand order), fused epilogues: rotary
//                         embedding of q,k / residual add / LayerNorm(192) + GELU
//   lg_attention_kernel   out = softmax(Q K^T) V, flash style: S never leaves registers; the softmax-ed tile IS the
//                         MFMA operand of the P.V product (key pairing (k, k+4)); key splits fill the chip
//   lg_dot_kernel         Linear(96 -> 1): matchability logits
//   lg_prune_*            ordered compaction of the rows whose matchability > 1 - width_confidence
//   lg_transpose_kernel   (N,96) -> (96,Npad) so that the similarity matrix is a plain row-major linear layer
//   lg_row_lse / lg_col_lse / lg_row_best / lg_col_best / lg_mutual   double log-softmax assignment + mutual filter
// Every per-set kernel processes BOTH images of the pair in one launch (blockIdx.y / .z = image).
#include "kernels.hpp"

namespace xfh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LG_D = 96;

__device__ inline int lg_live(const int32_t* n_dev, int cap) {
    const int v = n_dev ? *n_dev : cap;
    return v < 0 ? 0 : (v > cap ? cap : v);
}

// ------------------------------------------------------------------------------------------
// cos/sin tables: kn = (kp - size/2) / (max(W,H)/2); proj = Wr . kn (48); entries 2f, 2f+1 share frequency f
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lg_encode_kernel(const float* __restrict__ kpts, int N, float W, float H, const float* __restrict__ wr,
                                                        float* __restrict__ cs, float* __restrict__ sn) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int n = g / 48, f = g - n * 48;
    if (n >= N) return;
    const float sc = fmaxf(W, H) / 2;
    const float kx = (kpts[2 * n] - W / 2) / sc, ky = (kpts[2 * n + 1] - H / 2) / sc;
    const float p = kx * wr[2 * f] + ky * wr[2 * f + 1];
    const float c = cosf(p), s = sinf(p);
    cs[(size_t)n * LG_D + 2 * f] = c; cs[(size_t)n * LG_D + 2 * f + 1] = c;
    sn[(size_t)n * LG_D + 2 * f] = s; sn[(size_t)n * LG_D + 2 * f + 1] = s;
}
void launch_lg_encode(const float* kpts, int N, float W, float H, const float* wr, float* cs, float* sn, hipStream_t st) {
    lg_encode_kernel<<<ceil_div(N * 48, 256), 256, 0, st>>>(kpts, N, W, H, wr, cs, sn);
}

// ------------------------------------------------------------------------------------------
// y (n, N) = x (n, K) . W^T + b with N = 32 * (waves per workgroup).  One workgroup = one block of 32 rows, wave w =
// output features 32w..32w+31.  MFMA orientation D[i = feature][j = row]: A = W, B = x, both read as float4 in
// operand order -- lane (l31, half) walks k = half*K/2 .. +K/2-1 (the k-pair of a step is (k, k + K/2)); W is packed
// on the host as [feature block][half][K/8][32 lanes][4] so a wave reads 2 x 512 contiguous bytes per step.
// The accumulator lane then owns ONE row and 16 features (4 runs of 4): row-wise epilogues need no cross-lane work
// beyond one half swap:
//   LG_EPI_STORE     y = v
//   LG_EPI_RESIDUAL  y += v
//   LG_EPI_ROTARY    features < 192 (q | k of the fused [q|k|v] projection): x' = x cos + rotate_half(x) sin
//   LG_EPI_LNGELU    N = 192 (6 waves): LayerNorm over the row (two-pass, partial sums through LDS) + exact GELU
// ------------------------------------------------------------------------------------------
template <int K, int EPI>
__global__ __launch_bounds__(576) void lg_linear_kernel(const float* __restrict__ wp, const float* __restrict__ bias, LgLinSide sa, LgLinSide sb,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta) {
    const LgLinSide S = blockIdx.y ? sb : sa;
    const int n = lg_live(S.n, S.cap);
    const int row0 = blockIdx.x * 32;
    if (row0 >= n) return;                      // uniform per workgroup
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31, wave = threadIdx.x >> 6;
    const int row = row0 + l31;
    const float4* xs = reinterpret_cast<const float4*>(S.x + (size_t)min(row, n - 1) * S.ldx + half * (K / 2));
    const float4* ws = reinterpret_cast<const float4*>(wp) + (size_t)(wave * 2 + half) * (K / 8) * 32 + l31;
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 b4 = *reinterpret_cast<const float4*>(bias + wave * 32 + 8 * g + 4 * half);
        acc[4 * g] = b4.x; acc[4 * g + 1] = b4.y; acc[4 * g + 2] = b4.z; acc[4 * g + 3] = b4.w;
    }
#pragma unroll
    for (int j = 0; j < K / 8; ++j) {
        const float4 xv = xs[j];
        const float4 wv = ws[j * 32];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.x, xv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.y, xv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.z, xv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv.w, xv.w, acc, 0, 0, 0);
    }
    // this lane: row `row`, features wave*32 + 8g + 4*half + {0..3}, g = 0..3  (acc[4g + e])
    if (EPI == LG_EPI_LNGELU) {
        __shared__ float red[2][6][32];
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[r];
        sum += __shfl_xor(sum, 32, 64);
        if (half == 0) red[0][wave][l31] = sum;
        __syncthreads();
        float mean = 0.f;
#pragma unroll
        for (int w = 0; w < 6; ++w) mean += red[0][w][l31];
        mean *= (1.f / 192.f);
        float q = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = acc[r] - mean; q += d * d; }
        q += __shfl_xor(q, 32, 64);
        if (half == 0) red[1][wave][l31] = q;
        __syncthreads();
        float var = 0.f;
#pragma unroll
        for (int w = 0; w < 6; ++w) var += red[1][w][l31];
        const float rstd = 1.f / sqrtf(var * (1.f / 192.f) + 1e-5f);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = wave * 32 + 8 * g + 4 * half;
            const float4 ga = *reinterpret_cast<const float4*>(gamma + c), be = *reinterpret_cast<const float4*>(beta + c);
            const float gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = (acc[4 * g + e] - mean) * rstd * gg[e] + bb[e];
                acc[4 * g + e] = 0.5f * y * (1.f + erff(y * 0.70710678118654752440f));
            }
        }
    }
    if (row >= n) return;
    float* yr = S.y + (size_t)row * S.ldy + wave * 32 + 4 * half;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float4 v = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        if (EPI == LG_EPI_ROTARY && wave < 6) {
            const int f = (wave % 3) * 32 + 8 * g + 4 * half;
            const float4 c = *reinterpret_cast<const float4*>(S.cs + (size_t)row * LG_D + f);
            const float4 sn = *reinterpret_cast<const float4*>(S.sn + (size_t)row * LG_D + f);
            v = make_float4(v.x * c.x + (-v.y) * sn.x, v.y * c.y + v.x * sn.y, v.z * c.z + (-v.w) * sn.z, v.w * c.w + v.z * sn.w);
        }
        if (EPI == LG_EPI_RESIDUAL) {
            const float4 o = *reinterpret_cast<const float4*>(yr + 8 * g);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *reinterpret_cast<float4*>(yr + 8 * g) = v;
    }
}

int launch_lg_linear(const float* wp, const float* bias, int K, int N, int epi, const LgLinSide* sides, int nsides, const float* gamma,
                     const float* beta, hipStream_t st) {
    if (N % 32 || N > 288 || nsides < 1 || nsides > 2 || (epi == LG_EPI_LNGELU && N != 192)) return -1;
    const int cap = max(sides[0].cap, nsides > 1 ? sides[1].cap : 0);
    if (cap <= 0) return 0;
    const dim3 grid(ceil_div(cap, 32), nsides), block(N * 2);
    const LgLinSide a = sides[0], b = sides[nsides - 1];
#define XFH_LGL(KV, EV) lg_linear_kernel<KV, EV><<<grid, block, 0, st>>>(wp, bias, a, b, gamma, beta)
    if (K == 64 && epi == LG_EPI_STORE) { XFH_LGL(64, LG_EPI_STORE); return 0; }
    if (K == 96 && epi == LG_EPI_STORE) { XFH_LGL(96, LG_EPI_STORE); return 0; }
    if (K == 96 && epi == LG_EPI_ROTARY) { XFH_LGL(96, LG_EPI_ROTARY); return 0; }
    if (K == 192 && epi == LG_EPI_LNGELU) { XFH_LGL(192, LG_EPI_LNGELU); return 0; }
    if (K == 192 && epi == LG_EPI_RESIDUAL) { XFH_LGL(192, LG_EPI_RESIDUAL); return 0; }
#undef XFH_LGL
    return -1;
}

// ------------------------------------------------------------------------------------------
// Attention: O[q] = sum_k softmax_k(scale * Q[q].K[k]) V[k]      (d = 96, one head, fp32 on v_mfma_f32_32x32x2_f32)
//
// Grid (query blocks of 128, key splits).  A workgroup = 4 waves x 32 queries sweeping ITS slice of the key tiles
// (32 keys each, K and V staged in LDS, double buffered: the next tile's global loads fly during the MFMAs).
//   S tile  D[i = key][j = query] : A = K tile (ds_read_b128: lane half h owns features 48h..48h+47 -- the k-pair of
//           MFMA step s is (s, 48+s)), B = Q (48 stationary registers, pre-multiplied by scale*log2(e))
//   online softmax in base 2 per query: lane (query, half) holds 16 keys -> register reduction + one cross-half
//           shuffle for the max; the running sum stays per lane until the end; the O rescale is skipped while no
//           query of the wave raised its max (exact: the factor would be 1)
//   P.V     D[i = d][j = query]   : B = the p[r] registers (halves hold keys (k, k+4) = the k-pair of the step),
//           A = V read from LDS -- P never touches LDS
// Key splits (flash-decoding style) fill the chip when there are few query blocks: split s writes un-normalised
// (o, m, l) partials and lg_attention_combine_kernel folds them.  With one split the kernel writes O directly.
// ------------------------------------------------------------------------------------------
constexpr int LGA_KS = 100;      // K tile row stride (floats): == 4 mod 32 -> conflict-free ds_read_b128, 16-B aligned rows
constexpr int LGA_MAXSPLIT = 16;

// one launch serves both images of a pair, so a side aims at ~256 workgroups (the chip holds 512: 2 per CU)
int lg_attention_splits(int qcap, int kcap) {
    const int nqb = ceil_div(max(qcap, 1), 128);
    int ns = min(min(ceil_div(256, nqb), LGA_MAXSPLIT), max(kcap, 1) / 64);
    return max(ns, 1);
}
size_t lg_attention_partial_floats(int qcap, int kcap) {
    const int ns = lg_attention_splits(qcap, kcap);
    return ns > 1 ? (size_t)ns * qcap * (LG_D + 2) : 0;
}

__global__ __launch_bounds__(256, 2) void lg_attention_kernel(LgAttSide sa, LgAttSide sb, int ldq, int ldk, int ldv, int ldo, float scale_log2e) {
    const LgAttSide S = blockIdx.z ? sb : sa;
    const float* __restrict__ Q = S.Q;
    const float* __restrict__ K = S.K;
    const float* __restrict__ V = S.V;
    float* __restrict__ O = S.O;
    float* __restrict__ part = S.part;
    const int qcap = S.qcap, kcap = S.kcap;
    const int32_t *nq_dev = S.nq, *nk_dev = S.nk;
    __shared__ __attribute__((aligned(16))) float Kl[2][32 * LGA_KS];
    __shared__ __attribute__((aligned(16))) float Vl[2][32 * LG_D];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31, wave = tid >> 6;
    const int nq = lg_live(nq_dev, qcap), nk = lg_live(nk_dev, kcap);
    const int nsplit = S.nsplit, split = blockIdx.y;
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    if (blockIdx.x * 128 >= nq || split >= nsplit) return;          // uniform per workgroup
    const int ntile = (nk + 31) >> 5;
    const int t_begin = (int)((long)ntile * split / nsplit), t_end = (int)((long)ntile * (split + 1) / nsplit);

    float qf[48];
    {
        const int qrow = min(q0 + l31, nq - 1);
        const float4* src = reinterpret_cast<const float4*>(Q + (size_t)qrow * ldq + 48 * half);
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            const float4 v = src[c];
            qf[4 * c + 0] = v.x * scale_log2e; qf[4 * c + 1] = v.y * scale_log2e; qf[4 * c + 2] = v.z * scale_log2e; qf[4 * c + 3] = v.w * scale_log2e;
        }
    }
    f32x16 o[3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[m][r] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;          // lrun: this lane's share of the row sum (its 16 keys per tile)

    // staging: 32 rows x 24 float4 of K and of V per tile = 3 + 3 float4 per thread
    float4 kreg0, kreg1, kreg2, vreg0, vreg1, vreg2;      // (separate scalars: arrays captured by lambdas ended up in scratch)
    const int sr0 = tid / 24, sc0 = tid - sr0 * 24, sr1 = (tid + 256) / 24, sc1 = tid + 256 - sr1 * 24, sr2 = (tid + 512) / 24, sc2 = tid + 512 - sr2 * 24;
#define LGA_FETCH(t)                                                                                  \
    {                                                                                                 \
        const int kr0 = min((t) * 32 + sr0, nk - 1), kr1 = min((t) * 32 + sr1, nk - 1), kr2 = min((t) * 32 + sr2, nk - 1); \
        kreg0 = *reinterpret_cast<const float4*>(K + (size_t)kr0 * ldk + 4 * sc0);                    \
        kreg1 = *reinterpret_cast<const float4*>(K + (size_t)kr1 * ldk + 4 * sc1);                    \
        kreg2 = *reinterpret_cast<const float4*>(K + (size_t)kr2 * ldk + 4 * sc2);                    \
        vreg0 = *reinterpret_cast<const float4*>(V + (size_t)kr0 * ldv + 4 * sc0);                    \
        vreg1 = *reinterpret_cast<const float4*>(V + (size_t)kr1 * ldv + 4 * sc1);                    \
        vreg2 = *reinterpret_cast<const float4*>(V + (size_t)kr2 * ldv + 4 * sc2);                    \
    }
#define LGA_STASH(buf)                                                                                \
    {                                                                                                 \
        *reinterpret_cast<float4*>(&Kl[buf][sr0 * LGA_KS + 4 * sc0]) = kreg0;                         \
        *reinterpret_cast<float4*>(&Kl[buf][sr1 * LGA_KS + 4 * sc1]) = kreg1;                         \
        *reinterpret_cast<float4*>(&Kl[buf][sr2 * LGA_KS + 4 * sc2]) = kreg2;                         \
        *reinterpret_cast<float4*>(&Vl[buf][sr0 * LG_D + 4 * sc0]) = vreg0;                           \
        *reinterpret_cast<float4*>(&Vl[buf][sr1 * LG_D + 4 * sc1]) = vreg1;                           \
        *reinterpret_cast<float4*>(&Vl[buf][sr2 * LG_D + 4 * sc2]) = vreg2;                           \
    }
    kreg0 = kreg1 = kreg2 = vreg0 = vreg1 = vreg2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t_begin < t_end) {
        LGA_FETCH(t_begin);
        LGA_STASH(0);
    }
    __syncthreads();

    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        if (t + 1 < t_end) LGA_FETCH(t + 1);
        const float* kl = &Kl[buf][l31 * LGA_KS + 48 * half];
        const float* vl = &Vl[buf][(4 * half) * LG_D + l31];
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int c = 0; c < 12; ++c) {
            const float4 kv = *reinterpret_cast<const float4*>(kl + 4 * c);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.x, qf[4 * c + 0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.y, qf[4 * c + 1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.z, qf[4 * c + 2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv.w, qf[4 * c + 3], s, 0, 0, 0);
        }
        // this lane: query l31, keys 32t + (r&3) + 8(r>>2) + 4*half
        if (t * 32 + 32 > nk) {                  // ragged last tile (uniform branch)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half >= nk) s[r] = -INFINITY;
        }
        float tmax = s[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
        tmax = fmaxf(tmax, xhalf(tmax));
        const float mnew = fmaxf(mrun, tmax);     // finite: every tile holds at least one live key
        if (__any(mnew != mrun)) {
            const float corr = __builtin_amdgcn_exp2f(mrun - mnew);           // raw v_exp_f32; exp2(-inf) = 0 on the first tile
            lrun *= corr;
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[m][r] *= corr;
            mrun = mnew;
        }
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[r] = __builtin_amdgcn_exp2f(s[r] - mrun); psum += s[r]; }      // (exp2f() costs 6 VALU ops for the denormal range: p < 2^-126 may flush to 0 here)
        lrun += psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* vr = vl + ((r & 3) + 8 * (r >> 2)) * LG_D;
#pragma unroll
            for (int m = 0; m < 3; ++m) o[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[m * 32], s[r], o[m], 0, 0, 0);
        }
        if (t + 1 < t_end) LGA_STASH(buf ^ 1);
        __syncthreads();
    }
#undef LGA_FETCH
#undef LGA_STASH
    lrun += __shfl_xor(lrun, 32, 64);
    if (q0 + l31 >= nq) return;
    if (nsplit == 1) {
        const float inv = lrun > 0.f ? 1.f / lrun : 0.f;
        float* op = O + (size_t)(q0 + l31) * ldo;
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = make_float4(o[m][4 * g] * inv, o[m][4 * g + 1] * inv, o[m][4 * g + 2] * inv, o[m][4 * g + 3] * inv);
                *reinterpret_cast<float4*>(op + m * 32 + 8 * g + 4 * half) = v;
            }
    } else {
        float* po = part + ((size_t)split * qcap + q0 + l31) * (LG_D + 2);
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float* d = po + m * 32 + 8 * g + 4 * half;       // rows are 98 floats: 8-B aligned only
                *reinterpret_cast<float2*>(d) = make_float2(o[m][4 * g], o[m][4 * g + 1]);
                *reinterpret_cast<float2*>(d + 2) = make_float2(o[m][4 * g + 2], o[m][4 * g + 3]);
            }
        if (half == 0) *reinterpret_cast<float2*>(po + LG_D) = make_float2(mrun, lrun);
    }
}

// O[q] = sum_s 2^(m_s - M) o_s / sum_s 2^(m_s - M) l_s ; 32 lanes per query row (3 columns each)
__global__ __launch_bounds__(256) void lg_attention_combine_kernel(LgAttSide sa, LgAttSide sb, int ldo) {
    const LgAttSide S = blockIdx.y ? sb : sa;
    const int nsplit = S.nsplit, qcap = S.qcap;
    const float* __restrict__ part = S.part;
    const int q = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    if (nsplit <= 1 || q >= lg_live(S.nq, qcap)) return;
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, part[((size_t)s * qcap + q) * (LG_D + 2) + LG_D]);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, L = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float* p = part + ((size_t)s * qcap + q) * (LG_D + 2);
        const float ms = p[LG_D];
        if (ms == -INFINITY) continue;            // empty slice
        const float w = exp2f(ms - M);
        L += w * p[LG_D + 1];
        a0 += w * p[l]; a1 += w * p[l + 32]; a2 += w * p[l + 64];
    }
    const float inv = L > 0.f ? 1.f / L : 0.f;
    float* op = S.O + (size_t)q * ldo;
    op[l] = a0 * inv; op[l + 32] = a1 * inv; op[l + 64] = a2 * inv;
}

// sides[i].nsplit / .part are filled in here (part: consecutive slices of `scratch`, lg_attention_partial_floats each)
void launch_lg_attention(LgAttSide* sides, int nsides, int ldq, int ldk, int ldv, int ldo, float* scratch, float scale, hipStream_t st) {
    int qmax = 0, smax = 1;
    size_t off = 0;
    for (int i = 0; i < nsides; ++i) {
        if (sides[i].qcap <= 0 || sides[i].kcap <= 0) return;
        sides[i].nsplit = lg_attention_splits(sides[i].qcap, sides[i].kcap);
        sides[i].part = scratch + off;
        off += lg_attention_partial_floats(sides[i].qcap, sides[i].kcap);
        qmax = max(qmax, sides[i].qcap);
        smax = max(smax, sides[i].nsplit);
    }
    const LgAttSide a = sides[0], b = sides[nsides - 1];
    lg_attention_kernel<<<dim3(ceil_div(qmax, 128), smax, nsides), 256, 0, st>>>(a, b, ldq, ldk, ldv, ldo, scale * 1.44269504088896340736f);
    if (smax > 1) lg_attention_combine_kernel<<<dim3(ceil_div(qmax, 8), nsides), 256, 0, st>>>(a, b, ldo);
}

// z[n] = x[n] . w + b   (96 terms; 32 lanes x 3 per row, two rows per wave); blockIdx.y = image
__global__ __launch_bounds__(256) void lg_dot_kernel(LgRowSide sa, LgRowSide sb, int ld, const float* __restrict__ w, const float* __restrict__ b) {
    const LgRowSide S = blockIdx.y ? sb : sa;
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    const bool ok = row < lg_live(S.n, S.cap);
    float s = 0.f;
    if (ok) {
        const float* p = S.x + (size_t)row * ld;
        s = p[l] * w[l] + p[l + 32] * w[l + 32] + p[l + 64] * w[l + 64];
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (ok && l == 0) S.z[row] = s + b[0];
}
void launch_lg_dot(const LgRowSide* sides, int nsides, int ld, const float* w, const float* b, hipStream_t st) {
    const int cap = max(sides[0].cap, sides[nsides - 1].cap);
    if (cap <= 0) return;
    lg_dot_kernel<<<dim3(ceil_div(cap, 8), nsides), 256, 0, st>>>(sides[0], sides[nsides - 1], ld, w, b);
}

// ------------------------------------------------------------------------------------------
// width pruning: rows with sigmoid(z) > thr survive, order kept.  lg_prune_map builds the ordered list of
// surviving source rows (one 1024-thread workgroup per set) and the new live count; lg_gather_rows copies the
// descriptor / cos / sin rows and the original-index list into the other half of a ping-pong buffer pair.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void lg_prune_map_kernel(LgPruneSide sa, LgPruneSide sb, float thr, int min_kpts) {
    __shared__ int wsum[16];
    __shared__ int s_base;
    const LgPruneSide S = blockIdx.x ? sb : sa;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lg_live(S.n_in, S.cap);
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int row = base + tid;
        const bool keep = row < n && (n <= min_kpts || 1.f / (1.f + expf(-S.z[row])) > thr);      // a set is only pruned while it is larger than min_kpts
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int off = s_base, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int sw = wsum[w];
            if (w < wave) off += sw;
            tot += sw;
        }
        if (keep) S.map[off + before] = row;
        __syncthreads();
        if (tid == 0) s_base += tot;
        __syncthreads();
    }
    if (tid == 0) *S.n_out = s_base;
}
__global__ __launch_bounds__(256) void lg_gather_rows_kernel(LgPruneSide sa, LgPruneSide sb, int ldx) {
    const LgPruneSide S = blockIdx.y ? sb : sa;
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int n = g / 24, c4 = g - n * 24;
    if (n >= lg_live(S.n_out, S.cap)) return;
    const int src = S.map[n];
    *reinterpret_cast<float4*>(S.xo + (size_t)n * ldx + 4 * c4) = *reinterpret_cast<const float4*>(S.x + (size_t)src * ldx + 4 * c4);
    *reinterpret_cast<float4*>(S.cso + (size_t)n * LG_D + 4 * c4) = *reinterpret_cast<const float4*>(S.cs + (size_t)src * LG_D + 4 * c4);
    *reinterpret_cast<float4*>(S.sno + (size_t)n * LG_D + 4 * c4) = *reinterpret_cast<const float4*>(S.sn + (size_t)src * LG_D + 4 * c4);
    if (c4 == 0) S.indo[n] = S.ind[src];
}
void launch_lg_prune(const LgPruneSide* sides, int nsides, float thr, int min_kpts, int ldx, hipStream_t st) {
    const int cap = max(sides[0].cap, sides[nsides - 1].cap);
    if (cap <= 0) return;
    lg_prune_map_kernel<<<nsides, 1024, 0, st>>>(sides[0], sides[nsides - 1], thr, min_kpts);
    lg_gather_rows_kernel<<<dim3(ceil_div(cap * 24, 256), nsides), 256, 0, st>>>(sides[0], sides[nsides - 1], ldx);
}

// ------------------------------------------------------------------------------------------
// (N, 96) rows (leading dimension ld) -> (96, npad) with zeros beyond the live count: the right-hand side of
// sim = md0 . md1^T as a plain [K][n_pad] weight image for linear_mfma_kernel
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lg_transpose_kernel(const float* __restrict__ x, int ld, const int32_t* __restrict__ n_dev, int cap,
                                                           float* __restrict__ xt, int npad) {
    __shared__ float t[32][33];
    const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int n = lg_live(n_dev, cap);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = n0 + ty + 8 * i;
        t[ty + 8 * i][tx] = r < n ? x[(size_t)r * ld + k0 + tx] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + ty + 8 * i, c = n0 + tx;
        if (c < npad) xt[(size_t)k * npad + c] = t[tx][ty + 8 * i];
    }
}
void launch_lg_transpose(const float* x, int ld, const int32_t* n_dev, int cap, float* xt, int npad, hipStream_t st) {
    lg_transpose_kernel<<<dim3(ceil_div(npad, 32), LG_D / 32), 256, 0, st>>>(x, ld, n_dev, cap, xt, npad);
}

// ------------------------------------------------------------------------------------------
// Assignment on the materialised similarity matrix sim (n0 x n1, leading dimension ld):
//   score[i][j] = (sim - rowlse[i]) + (sim - collse[j]) + (logsigmoid(z0[i]) + logsigmoid(z1[j]))
// ------------------------------------------------------------------------------------------
__device__ inline float lg_logsigmoid(float z) { return fminf(z, 0.f) - log1pf(expf(-fabsf(z))); }

// one wave per row, float4 per lane (ld is a multiple of 64 floats: rows are 16-byte aligned)
__global__ __launch_bounds__(256) void lg_row_lse_kernel(const float* __restrict__ sim, int ld, const int32_t* __restrict__ n0_dev, int cap0,
                                                         const int32_t* __restrict__ n1_dev, int cap1, float* __restrict__ lse) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n0 = lg_live(n0_dev, cap0), n1 = lg_live(n1_dev, cap1);
    if (row >= n0) return;
    const float* p = sim + (size_t)row * ld;
    float m = -INFINITY;
    for (int j = 4 * lane; j < n1; j += 256) {
        const float4 v = *reinterpret_cast<const float4*>(p + j);
        m = fmaxf(m, v.x);
        if (j + 1 < n1) m = fmaxf(m, v.y);
        if (j + 2 < n1) m = fmaxf(m, v.z);
        if (j + 3 < n1) m = fmaxf(m, v.w);
    }
    m = wave_max(m);
    float s = 0.f;
    for (int j = 4 * lane; j < n1; j += 256) {
        const float4 v = *reinterpret_cast<const float4*>(p + j);
        s += expf(v.x - m);
        if (j + 1 < n1) s += expf(v.y - m);
        if (j + 2 < n1) s += expf(v.z - m);
        if (j + 3 < n1) s += expf(v.w - m);
    }
    s = wave_sum(s);
    if (lane == 0) lse[row] = m + logf(s);
}
// column statistics: 64 columns x one slice of the rows per workgroup (grid.y = LG_RSPLIT slices, the 4 waves of a
// workgroup interleave the rows of the slice: 256-byte coalesced reads); a tiny second kernel folds the slices
constexpr int LG_RSPLIT = 16;
__global__ __launch_bounds__(256) void lg_col_lse_part_kernel(const float* __restrict__ sim, int ld, const int32_t* __restrict__ n0_dev, int cap0,
                                                              const int32_t* __restrict__ n1_dev, int cap1, float2* __restrict__ part, int npad) {
    __shared__ float sm[4][64], ss[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int n0 = lg_live(n0_dev, cap0), n1 = lg_live(n1_dev, cap1);
    const int chunk = ceil_div(n0, LG_RSPLIT), r0 = blockIdx.y * chunk, r1 = min(n0, r0 + chunk);
    float m = -INFINITY, s = 0.f;
    if (col < n1)
        for (int i = r0 + wave; i < r1; i += 4) {
            const float v = sim[(size_t)i * ld + col];
            const float mn = fmaxf(m, v);
            s = s * expf(m - mn) + expf(v - mn);
            m = mn;
        }
    sm[wave][lane] = m; ss[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && col < n1) {
        const float mm = fmaxf(fmaxf(sm[0][lane], sm[1][lane]), fmaxf(sm[2][lane], sm[3][lane]));
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) t += ss[w][lane] > 0.f ? ss[w][lane] * expf(sm[w][lane] - mm) : 0.f;
        part[(size_t)blockIdx.y * npad + col] = make_float2(mm, t);
    }
}
__global__ __launch_bounds__(256) void lg_col_lse_final_kernel(const float2* __restrict__ part, int npad, const int32_t* __restrict__ n1_dev, int cap1,
                                                               float* __restrict__ lse) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= lg_live(n1_dev, cap1)) return;
    float2 p[LG_RSPLIT];
    float mm = -INFINITY;
#pragma unroll
    for (int k = 0; k < LG_RSPLIT; ++k) { p[k] = part[(size_t)k * npad + col]; mm = fmaxf(mm, p[k].x); }
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < LG_RSPLIT; ++k) t += p[k].y > 0.f ? p[k].y * expf(p[k].x - mm) : 0.f;
    lse[col] = mm + logf(t);
}
// z -> logsigmoid(z) in place, once per key-point (the score kernels would otherwise evaluate it per matrix element)
__global__ __launch_bounds__(256) void lg_logsigmoid_kernel(float* __restrict__ z0, const int32_t* __restrict__ n0_dev, int cap0, float* __restrict__ z1,
                                                           const int32_t* __restrict__ n1_dev, int cap1) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    float* z = blockIdx.y ? z1 : z0;
    if (g < (blockIdx.y ? lg_live(n1_dev, cap1) : lg_live(n0_dev, cap0))) z[g] = lg_logsigmoid(z[g]);
}
__device__ inline float lg_score(float v, float rl, float cl, float a, float b) { return ((v - rl) + (v - cl)) + (a + b); }
// row arg-max (first index on ties) of the core scores; one wave per row, float4 per lane
__global__ __launch_bounds__(256) void lg_row_best_kernel(const float* __restrict__ sim, int ld, const int32_t* __restrict__ n0_dev, int cap0,
                                                          const int32_t* __restrict__ n1_dev, int cap1, const float* __restrict__ rlse,
                                                          const float* __restrict__ clse, const float* __restrict__ z0,
                                                          const float* __restrict__ z1, int32_t* __restrict__ m0, float* __restrict__ best0) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n0 = lg_live(n0_dev, cap0), n1 = lg_live(n1_dev, cap1);
    if (row >= n0) return;
    const float* p = sim + (size_t)row * ld;
    const float rl = rlse[row], a = z0[row];
    float bv = -INFINITY;
    int bj = 0x7fffffff;
    for (int j = 4 * lane; j < n1; j += 256) {           // ascending j within the lane: strict > keeps the first maximum
        const float4 sv = *reinterpret_cast<const float4*>(p + j);
        const float4 cl = *reinterpret_cast<const float4*>(clse + j);      // (clse / z1 hold n1pad entries; the tail is masked below)
        const float4 zb = *reinterpret_cast<const float4*>(z1 + j);
        const float s4[4] = {sv.x, sv.y, sv.z, sv.w}, c4[4] = {cl.x, cl.y, cl.z, cl.w}, b4[4] = {zb.x, zb.y, zb.z, zb.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = lg_score(s4[e], rl, c4[e], a, b4[e]);
            if (j + e < n1 && v > bv) { bv = v; bj = j + e; }
        }
    }
    unsigned long long key = bj == 0x7fffffff ? 0ull : (((unsigned long long)float_ord(bv) << 32) | (0xffffffffu - (unsigned)bj));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) key = u64_max(key, shfl_xor_u64(key, o));
    if (lane == 0) {
        m0[row] = key ? (int)(0xffffffffu - (unsigned)(key & 0xffffffffu)) : -1;
        best0[row] = key ? ord_float((unsigned)(key >> 32)) : -INFINITY;
    }
}
// column arg-max (first row on ties): same slicing as the column LSE, packed keys (ord(score) << 32 | ~row)
__global__ __launch_bounds__(256) void lg_col_best_part_kernel(const float* __restrict__ sim, int ld, const int32_t* __restrict__ n0_dev, int cap0,
                                                               const int32_t* __restrict__ n1_dev, int cap1, const float* __restrict__ rlse,
                                                               const float* __restrict__ clse, const float* __restrict__ z0,
                                                               const float* __restrict__ z1, unsigned long long* __restrict__ part, int npad) {
    __shared__ unsigned long long sk[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int n0 = lg_live(n0_dev, cap0), n1 = lg_live(n1_dev, cap1);
    const int chunk = ceil_div(n0, LG_RSPLIT), r0 = blockIdx.y * chunk, r1 = min(n0, r0 + chunk);
    unsigned long long key = 0ull;
    if (col < n1) {
        const float cl = clse[col], b = z1[col];
        for (int i = r0 + wave; i < r1; i += 4) {
            const float v = lg_score(sim[(size_t)i * ld + col], rlse[i], cl, z0[i], b);
            key = u64_max(key, ((unsigned long long)float_ord(v) << 32) | (0xffffffffu - (unsigned)i));
        }
    }
    sk[wave][lane] = key;
    __syncthreads();
    if (wave == 0 && col < n1)
        part[(size_t)blockIdx.y * npad + col] = u64_max(u64_max(sk[0][lane], sk[1][lane]), u64_max(sk[2][lane], sk[3][lane]));
}
__global__ __launch_bounds__(256) void lg_col_best_final_kernel(const unsigned long long* __restrict__ part, int npad, const int32_t* __restrict__ n1_dev,
                                                                int cap1, int32_t* __restrict__ m1) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= lg_live(n1_dev, cap1)) return;
    unsigned long long key = 0ull;
#pragma unroll
    for (int k = 0; k < LG_RSPLIT; ++k) key = u64_max(key, part[(size_t)k * npad + col]);
    m1[col] = key ? (int)(0xffffffffu - (unsigned)(key & 0xffffffffu)) : -1;
}
// mutual check + threshold, ordered output (ascending image-0 index): matches (S,2) as ORIGINAL indices, scores (S)
__global__ __launch_bounds__(1024) void lg_mutual_kernel(const int32_t* __restrict__ m0, const int32_t* __restrict__ m1, const float* __restrict__ best0,
                                                         const int32_t* __restrict__ ind0, const int32_t* __restrict__ ind1,
                                                         const int32_t* __restrict__ n0_dev, int cap0, float thr, int64_t* __restrict__ matches,
                                                         float* __restrict__ scores, int32_t* __restrict__ n_out) {
    __shared__ int wsum[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = lg_live(n0_dev, cap0);
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int base = 0; base < n0; base += 1024) {
        const int i = base + tid;
        bool keep = false;
        int j = -1;
        float sc = 0.f;
        if (i < n0) {
            j = m0[i];
            if (j >= 0 && m1[j] == i) {
                sc = expf(best0[i]);
                keep = sc > thr;
            }
        }
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int off = s_base, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int sw = wsum[w];
            if (w < wave) off += sw;
            tot += sw;
        }
        if (keep) {
            matches[2 * (size_t)(off + before)] = ind0[i];
            matches[2 * (size_t)(off + before) + 1] = ind1[j];
            scores[off + before] = sc;
        }
        __syncthreads();
        if (tid == 0) s_base += tot;
        __syncthreads();
    }
    if (tid == 0) *n_out = s_base;
}
size_t lg_assign_scratch_bytes(int cap1) { return (size_t)LG_RSPLIT * ((cap1 + 63) / 64 * 64) * 8; }
// z0 / z1: matchability logits, overwritten with their logsigmoid.
// `scratch`: lg_assign_scratch_bytes(cap1) bytes (8-byte aligned), used for the column partials of both passes
void launch_lg_assign(const float* sim, int ld, const int32_t* n0_dev, int cap0, const int32_t* n1_dev, int cap1, float* z0, float* z1,
                      float* rlse, float* clse, int32_t* m0, int32_t* m1, float* best0, const int32_t* ind0, const int32_t* ind1, float thr,
                      int64_t* matches, float* scores, int32_t* n_out, void* scratch, hipStream_t st) {
    const int npad = (cap1 + 63) / 64 * 64;
    lg_logsigmoid_kernel<<<dim3(ceil_div(max(cap0, cap1), 256), 2), 256, 0, st>>>(z0, n0_dev, cap0, z1, n1_dev, cap1);
    lg_row_lse_kernel<<<ceil_div(cap0, 4), 256, 0, st>>>(sim, ld, n0_dev, cap0, n1_dev, cap1, rlse);
    lg_col_lse_part_kernel<<<dim3(npad / 64, LG_RSPLIT), 256, 0, st>>>(sim, ld, n0_dev, cap0, n1_dev, cap1, (float2*)scratch, npad);
    lg_col_lse_final_kernel<<<ceil_div(cap1, 256), 256, 0, st>>>((const float2*)scratch, npad, n1_dev, cap1, clse);
    lg_row_best_kernel<<<ceil_div(cap0, 4), 256, 0, st>>>(sim, ld, n0_dev, cap0, n1_dev, cap1, rlse, clse, z0, z1, m0, best0);
    lg_col_best_part_kernel<<<dim3(npad / 64, LG_RSPLIT), 256, 0, st>>>(sim, ld, n0_dev, cap0, n1_dev, cap1, rlse, clse, z0, z1,
                                                                        (unsigned long long*)scratch, npad);
    lg_col_best_final_kernel<<<ceil_div(cap1, 256), 256, 0, st>>>((const unsigned long long*)scratch, npad, n1_dev, cap1, m1);
    lg_mutual_kernel<<<1, 1024, 0, st>>>(m0, m1, best0, ind0, ind1, n0_dev, cap0, thr, matches, scores, n_out);
}

}  // namespace xfh
