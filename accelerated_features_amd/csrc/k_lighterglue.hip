// LighterGlue (kornia LightGlue under modules/lighterglue.py:12-27: d = 96, one head, 6 layers) -- device kernels.
// Call sites in the reference: modules/lighterglue.py:50-57, modules/xfeat.py:131-162.  Arithmetic: the published
// LightGlue v0.1 algorithm (kornia 0.7.2 is not on disk; see oracle/lighterglue_oracle.py for what is pinned).
//
// One pair per call (the reference supports B = 1 only).  Everything keeps a fixed capacity N and a device-side
// live count (width pruning shrinks the sets after every layer): no host read-back until the final match list.
//
//   lg_encode_kernel      key-point normalisation + learnable Fourier encoding -> cos / sin tables (N, 96)
//   lg_rotary_kernel      rotary embedding of q and k in place (pairs (2i, 2i+1))
//   lg_attention_kernel   out = softmax(Q K^T) V, flash style on v_mfma_f32_32x32x2_f32: S never leaves registers;
//                         the softmax-ed tile IS the MFMA operand of the P.V product (key pairing (k, k+4))
//   lg_ln_gelu_kernel     LayerNorm(192) + exact GELU, in place
//   lg_add_kernel         residual add
//   lg_dot_kernel         Linear(96 -> 1) (+ sigmoid): matchability / token heads
//   lg_prune_kernel       ordered compaction of the rows whose matchability > 1 - width_confidence
//   lg_transpose_kernel   (N,96) -> (96,Npad) so that a similarity matrix is a plain row-major linear layer
//   lg_row_lse / lg_col_lse / lg_row_best / lg_col_best / lg_mutual   double log-softmax assignment + mutual filter
// Linear layers reuse linear_mfma_kernel (k_linear_mfma.hip).
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

// x' = x*cos + rotate_half(x)*sin on q (cols 0..95) and k (cols 96..191) of the fused projection buffer
__global__ __launch_bounds__(256) void lg_rotary_kernel(float* __restrict__ qkv, int ld, const int32_t* __restrict__ n_dev, int cap,
                                                        const float* __restrict__ cs, const float* __restrict__ sn) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int n = g / 96, j = g - n * 96;             // j: pair index over q (0..47) and k (48..95)
    if (n >= lg_live(n_dev, cap)) return;
    const int which = j / 48, f = j - which * 48;
    float* p = qkv + (size_t)n * ld + which * LG_D + 2 * f;
    const float x0 = p[0], x1 = p[1];
    const float c0 = cs[(size_t)n * LG_D + 2 * f], c1 = cs[(size_t)n * LG_D + 2 * f + 1];
    const float s0 = sn[(size_t)n * LG_D + 2 * f], s1 = sn[(size_t)n * LG_D + 2 * f + 1];
    p[0] = x0 * c0 + (-x1) * s0;
    p[1] = x1 * c1 + x0 * s1;
}
void launch_lg_rotary(float* qkv, int ld, const int32_t* n_dev, int cap, const float* cs, const float* sn, hipStream_t st) {
    lg_rotary_kernel<<<ceil_div(cap * 96, 256), 256, 0, st>>>(qkv, ld, n_dev, cap, cs, sn);
}

// ------------------------------------------------------------------------------------------
// Attention: O[q] = sum_k softmax_k(scale * Q[q].K[k]) V[k]      (d = 96, one head)
// Workgroup = 4 waves x 32 queries; key/value tiles of 32 rows staged in LDS and shared by the waves.
//   S tile  D[i = key][j = query] : A = K tile (LDS), B = Q (48 stationary registers, pre-scaled)
//   online softmax per query: lane (query, half) holds 16 keys -> 16-way register reduction + one shuffle
//   P.V     D[i = d][j = query]   : B = the p[r] registers (lanes half 0 / 1 hold keys (k, k+4): the k-pair of an
//           MFMA step), A = V^T read from LDS -- no LDS round trip for P
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lg_attention_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ K, int ldk,
                                                           const float* __restrict__ V, int ldv, float* __restrict__ O, int ldo,
                                                           const int32_t* __restrict__ nq_dev, const int32_t* __restrict__ nk_dev, int qcap,
                                                           int kcap, float scale) {
    constexpr int KS = 97;                        // odd row stride: conflict-free column reads of the K tile
    __shared__ float Kl[32 * KS];
    __shared__ float Vl[32 * LG_D];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31, wave = tid >> 6;
    const int nq = lg_live(nq_dev, qcap), nk = lg_live(nk_dev, kcap);
    const int q0 = (blockIdx.x * 4 + wave) * 32;
    if (blockIdx.x * 128 >= nq) return;          // uniform per workgroup
    const int qrow = min(q0 + l31, max(nq - 1, 0));
    float qf[48];
#pragma unroll
    for (int s = 0; s < 48; ++s) qf[s] = Q[(size_t)qrow * ldq + 2 * s + half] * scale;
    f32x16 o[3];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[m][r] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;

    for (int k0 = 0; k0 < nk; k0 += 32) {
        __syncthreads();
        for (int e = tid; e < 32 * 24; e += 256) {          // 32 rows x 24 float4
            const int r = e / 24, c4 = e - r * 24;
            const int kr = min(k0 + r, nk - 1);
            const float4 kv = *reinterpret_cast<const float4*>(K + (size_t)kr * ldk + 4 * c4);
            const float4 vv = *reinterpret_cast<const float4*>(V + (size_t)kr * ldv + 4 * c4);
            float* kd = Kl + r * KS + 4 * c4;
            kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
            *reinterpret_cast<float4*>(Vl + r * LG_D + 4 * c4) = vv;
        }
        __syncthreads();
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
        for (int st = 0; st < 48; ++st) s = __builtin_amdgcn_mfma_f32_32x32x2f32(Kl[l31 * KS + 2 * st + half], qf[st], s, 0, 0, 0);
        // this lane: query l31, keys k0 + (r&3) + 8(r>>2) + 4*half
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (key >= nk) s[r] = -INFINITY;
            tmax = fmaxf(tmax, s[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mnew = fmaxf(mrun, tmax);
        const float corr = expf(mrun - mnew);               // exp(-inf) = 0 on the first tile
        float psum = 0.f;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { p[r] = expf(s[r] - mnew); psum += p[r]; }
        psum += __shfl_xor(psum, 32, 64);
        lrun = lrun * corr + psum;
        mrun = mnew;
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[m][r] *= corr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
            for (int m = 0; m < 3; ++m) o[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vl[key * LG_D + m * 32 + l31], p[r], o[m], 0, 0, 0);
        }
    }
    if (q0 + l31 < nq) {
        const float inv = lrun > 0.f ? 1.f / lrun : 0.f;
        float* op = O + (size_t)(q0 + l31) * ldo;
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) op[m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = o[m][r] * inv;
    }
}
void launch_lg_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* O, int ldo, const int32_t* nq_dev,
                         const int32_t* nk_dev, int qcap, int kcap, float scale, hipStream_t st) {
    if (qcap <= 0 || kcap <= 0) return;
    lg_attention_kernel<<<ceil_div(qcap, 128), 256, 0, st>>>(Q, ldq, K, ldk, V, ldv, O, ldo, nq_dev, nk_dev, qcap, kcap, scale);
}

// ------------------------------------------------------------------------------------------
// LayerNorm(192, eps 1e-5, affine) + exact GELU, in place; one wave per row (3 values per lane)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lg_ln_gelu_kernel(float* __restrict__ x, int ld, const int32_t* __restrict__ n_dev, int cap,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= lg_live(n_dev, cap)) return;
    float* p = x + (size_t)row * ld;
    float v[3];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { v[i] = p[lane + 64 * i]; s += v[i]; }
    const float mean = wave_sum(s) / 192.f;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { const float d = v[i] - mean; q += d * d; }
    const float rstd = 1.f / sqrtf(wave_sum(q) / 192.f + 1e-5f);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = lane + 64 * i;
        const float y = (v[i] - mean) * rstd * gamma[c] + beta[c];
        p[c] = 0.5f * y * (1.f + erff(y * 0.70710678118654752440f));
    }
}
void launch_lg_ln_gelu(float* x, int ld, const int32_t* n_dev, int cap, const float* gamma, const float* beta, hipStream_t st) {
    lg_ln_gelu_kernel<<<ceil_div(cap, 4), 256, 0, st>>>(x, ld, n_dev, cap, gamma, beta);
}

__global__ __launch_bounds__(256) void lg_add_kernel(float* __restrict__ x, int ldx, const float* __restrict__ y, int ldy,
                                                     const int32_t* __restrict__ n_dev, int cap) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int n = g / 24, c4 = g - n * 24;
    if (n >= lg_live(n_dev, cap)) return;
    float4 a = *reinterpret_cast<float4*>(x + (size_t)n * ldx + 4 * c4);
    const float4 b = *reinterpret_cast<const float4*>(y + (size_t)n * ldy + 4 * c4);
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    *reinterpret_cast<float4*>(x + (size_t)n * ldx + 4 * c4) = a;
}
void launch_lg_add(float* x, int ldx, const float* y, int ldy, const int32_t* n_dev, int cap, hipStream_t st) {
    lg_add_kernel<<<ceil_div(cap * 24, 256), 256, 0, st>>>(x, ldx, y, ldy, n_dev, cap);
}

// z[n] = x[n] . w + b   (96 terms; 32 lanes x 3 per row, two rows per wave)
__global__ __launch_bounds__(256) void lg_dot_kernel(const float* __restrict__ x, int ld, const int32_t* __restrict__ n_dev, int cap,
                                                     const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ z) {
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    const bool ok = row < lg_live(n_dev, cap);
    float s = 0.f;
    if (ok) {
        const float* p = x + (size_t)row * ld;
        s = p[l] * w[l] + p[l + 32] * w[l + 32] + p[l + 64] * w[l + 64];
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (ok && l == 0) z[row] = s + b[0];
}
void launch_lg_dot(const float* x, int ld, const int32_t* n_dev, int cap, const float* w, const float* b, float* z, hipStream_t st) {
    lg_dot_kernel<<<ceil_div(cap, 8), 256, 0, st>>>(x, ld, n_dev, cap, w, b, z);
}

// ------------------------------------------------------------------------------------------
// width pruning: rows with sigmoid(z) > thr survive, order kept.  lg_prune_map builds the ordered list of
// surviving source rows (one 1024-thread workgroup per set) and the new live count; lg_gather_rows copies the
// descriptor / cos / sin rows and the original-index list into the other half of a ping-pong buffer pair.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void lg_prune_map_kernel(const float* __restrict__ z, float thr, int min_kpts, const int32_t* __restrict__ n_in,
                                                            int cap, int32_t* __restrict__ map, int32_t* __restrict__ n_out) {
    __shared__ int wsum[16];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lg_live(n_in, cap);
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int row = base + tid;
        const bool keep = row < n && (n <= min_kpts || 1.f / (1.f + expf(-z[row])) > thr);      // a set is only pruned while it is larger than min_kpts
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
        if (keep) map[off + before] = row;
        __syncthreads();
        if (tid == 0) s_base += tot;
        __syncthreads();
    }
    if (tid == 0) *n_out = s_base;
}
__global__ __launch_bounds__(256) void lg_gather_rows_kernel(const int32_t* __restrict__ map, const int32_t* __restrict__ n_dev, int cap,
                                                             const float* __restrict__ x, int ldx, float* __restrict__ xo,
                                                             const float* __restrict__ cs, float* __restrict__ cso,
                                                             const float* __restrict__ sn, float* __restrict__ sno,
                                                             const int32_t* __restrict__ ind, int32_t* __restrict__ indo) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int n = g / 24, c4 = g - n * 24;
    if (n >= lg_live(n_dev, cap)) return;
    const int src = map[n];
    *reinterpret_cast<float4*>(xo + (size_t)n * ldx + 4 * c4) = *reinterpret_cast<const float4*>(x + (size_t)src * ldx + 4 * c4);
    *reinterpret_cast<float4*>(cso + (size_t)n * LG_D + 4 * c4) = *reinterpret_cast<const float4*>(cs + (size_t)src * LG_D + 4 * c4);
    *reinterpret_cast<float4*>(sno + (size_t)n * LG_D + 4 * c4) = *reinterpret_cast<const float4*>(sn + (size_t)src * LG_D + 4 * c4);
    if (c4 == 0) indo[n] = ind[src];
}
void launch_lg_prune(const float* z, float thr, int min_kpts, const int32_t* n_in, int cap, int32_t* map, int32_t* n_out, const float* x, int ldx, float* xo,
                     const float* cs, float* cso, const float* sn, float* sno, const int32_t* ind, int32_t* indo, hipStream_t st) {
    lg_prune_map_kernel<<<1, 1024, 0, st>>>(z, thr, min_kpts, n_in, cap, map, n_out);
    lg_gather_rows_kernel<<<ceil_div(cap * 24, 256), 256, 0, st>>>(map, n_out, cap, x, ldx, xo, cs, cso, sn, sno, ind, indo);
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

// one wave per row
__global__ __launch_bounds__(256) void lg_row_lse_kernel(const float* __restrict__ sim, int ld, const int32_t* __restrict__ n0_dev, int cap0,
                                                         const int32_t* __restrict__ n1_dev, int cap1, float* __restrict__ lse) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n0 = lg_live(n0_dev, cap0), n1 = lg_live(n1_dev, cap1);
    if (row >= n0) return;
    const float* p = sim + (size_t)row * ld;
    float m = -INFINITY;
    for (int j = lane; j < n1; j += 64) m = fmaxf(m, p[j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < n1; j += 64) s += expf(p[j] - m);
    s = wave_sum(s);
    if (lane == 0) lse[row] = m + logf(s);
}
// 64 columns per workgroup, the 4 waves split the rows
__global__ __launch_bounds__(256) void lg_col_lse_kernel(const float* __restrict__ sim, int ld, const int32_t* __restrict__ n0_dev, int cap0,
                                                         const int32_t* __restrict__ n1_dev, int cap1, float* __restrict__ lse) {
    __shared__ float sm[4][64], ss[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int n0 = lg_live(n0_dev, cap0), n1 = lg_live(n1_dev, cap1);
    float m = -INFINITY, s = 0.f;
    if (col < n1)
        for (int i = wave; i < n0; i += 4) {
            const float v = sim[(size_t)i * ld + col];
            const float mn = fmaxf(m, v);
            s = s * expf(m - mn) + expf(v - mn);
            m = mn;
        }
    sm[wave][lane] = m; ss[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && col < n1) {
        float mm = fmaxf(fmaxf(sm[0][lane], sm[1][lane]), fmaxf(sm[2][lane], sm[3][lane]));
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) t += ss[w][lane] > 0.f ? ss[w][lane] * expf(sm[w][lane] - mm) : 0.f;
        lse[col] = mm + logf(t);
    }
}
__device__ inline float lg_score(float v, float rl, float cl, float a, float b) { return ((v - rl) + (v - cl)) + (a + b); }
// row arg-max (first index on ties) of the core scores; one wave per row
__global__ __launch_bounds__(256) void lg_row_best_kernel(const float* __restrict__ sim, int ld, const int32_t* __restrict__ n0_dev, int cap0,
                                                          const int32_t* __restrict__ n1_dev, int cap1, const float* __restrict__ rlse,
                                                          const float* __restrict__ clse, const float* __restrict__ z0,
                                                          const float* __restrict__ z1, int32_t* __restrict__ m0, float* __restrict__ best0) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n0 = lg_live(n0_dev, cap0), n1 = lg_live(n1_dev, cap1);
    if (row >= n0) return;
    const float* p = sim + (size_t)row * ld;
    const float rl = rlse[row], a = lg_logsigmoid(z0[row]);
    float bv = -INFINITY;
    int bj = 0x7fffffff;
    for (int j = lane; j < n1; j += 64) {
        const float v = lg_score(p[j], rl, clse[j], a, lg_logsigmoid(z1[j]));
        if (v > bv) { bv = v; bj = j; }
    }
    unsigned long long key = bj == 0x7fffffff ? 0ull : (((unsigned long long)float_ord(bv) << 32) | (0xffffffffu - (unsigned)bj));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) key = u64_max(key, shfl_xor_u64(key, o));
    if (lane == 0) {
        m0[row] = key ? (int)(0xffffffffu - (unsigned)(key & 0xffffffffu)) : -1;
        best0[row] = key ? ord_float((unsigned)(key >> 32)) : -INFINITY;
    }
}
// column arg-max (first row on ties); 64 columns per workgroup, 4 waves split the rows
__global__ __launch_bounds__(256) void lg_col_best_kernel(const float* __restrict__ sim, int ld, const int32_t* __restrict__ n0_dev, int cap0,
                                                          const int32_t* __restrict__ n1_dev, int cap1, const float* __restrict__ rlse,
                                                          const float* __restrict__ clse, const float* __restrict__ z0,
                                                          const float* __restrict__ z1, int32_t* __restrict__ m1) {
    __shared__ unsigned long long sk[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int n0 = lg_live(n0_dev, cap0), n1 = lg_live(n1_dev, cap1);
    unsigned long long key = 0ull;
    if (col < n1) {
        const float cl = clse[col], b = lg_logsigmoid(z1[col]);
        for (int i = wave; i < n0; i += 4) {
            const float v = lg_score(sim[(size_t)i * ld + col], rlse[i], cl, lg_logsigmoid(z0[i]), b);
            key = u64_max(key, ((unsigned long long)float_ord(v) << 32) | (0xffffffffu - (unsigned)i));
        }
    }
    sk[wave][lane] = key;
    __syncthreads();
    if (wave == 0 && col < n1) {
        key = u64_max(u64_max(sk[0][lane], sk[1][lane]), u64_max(sk[2][lane], sk[3][lane]));
        m1[col] = key ? (int)(0xffffffffu - (unsigned)(key & 0xffffffffu)) : -1;
    }
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
void launch_lg_assign(const float* sim, int ld, const int32_t* n0_dev, int cap0, const int32_t* n1_dev, int cap1, const float* z0, const float* z1,
                      float* rlse, float* clse, int32_t* m0, int32_t* m1, float* best0, const int32_t* ind0, const int32_t* ind1, float thr,
                      int64_t* matches, float* scores, int32_t* n_out, hipStream_t st) {
    lg_row_lse_kernel<<<ceil_div(cap0, 4), 256, 0, st>>>(sim, ld, n0_dev, cap0, n1_dev, cap1, rlse);
    lg_col_lse_kernel<<<ceil_div(cap1, 64), 256, 0, st>>>(sim, ld, n0_dev, cap0, n1_dev, cap1, clse);
    lg_row_best_kernel<<<ceil_div(cap0, 4), 256, 0, st>>>(sim, ld, n0_dev, cap0, n1_dev, cap1, rlse, clse, z0, z1, m0, best0);
    lg_col_best_kernel<<<ceil_div(cap1, 64), 256, 0, st>>>(sim, ld, n0_dev, cap0, n1_dev, cap1, rlse, clse, z0, z1, m1);
    lg_mutual_kernel<<<1, 1024, 0, st>>>(m0, m1, best0, ind0, ind1, n0_dev, cap0, thr, matches, scores, n_out);
}

}  // namespace xfh
