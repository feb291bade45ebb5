// Mutual-nearest-neighbour matching, filter-and-refine form (the shipped path of xfh_match_mnn).
//   XFeat.match        modules/xfeat.py:327-348     XFeat.batch_match  modules/xfeat.py:265-290
//
// The exact kernel (k_match.hip) spends 32 f32 MFMAs (2048 pipe cycles) on every 32x32 tile of S = D1.D2^T to learn two things
// per row / column: the arg-max.  Here the same tile costs 4 bf16 MFMAs (128 cycles) -- as a FILTER -- and the decisions are
// still taken on exact fp32 numbers:
//   prep      D1, D2 -> bf16 copies (round-to-nearest-even), fp32 norms, per-pair maximum norm
//   sweep 1   S^ = D1^.D2^T on v_mfma_f32_32x32x16_bf16 (exact bf16 products, fp32 accumulation): row maxima, column maxima
//   sweep 2   the same S^ again (bit-identical): every (i,j) with  S^_ij >= rowmax^_i - 2 E_i   or   S^_ij >= colmax^_j - 2 E'_j
//             is appended to the pair's candidate list.   E_i = c |d1_i| max_j |d2_j|, E'_j = c |d2_j| max_i |d1_i|, c = 1.05 * 2^-8:
//             |S_ij - S^_ij| <= u (2 + u) sum_k |a_k b_k| + accumulation <= 2^-8 (1 + 2^-10) |a| |b| + 4e-6 |a| |b|   (u = 2^-9, RNE),
//             so the true arg-max of row i (and every exact tie with it) satisfies S^ >= S - E >= S_best^ ... >= rowmax^_i - 2 E_i:
//             the candidate list CONTAINS the exact arg-max of every row and of every column.  ~1.3 candidates per row.
//   exact     fp32 dot product of every candidate; row keys (ord(S) << 32 | ~j) and column keys (ord(S) << 32 | ~i) folded with
//             64-bit atomic max: the exact arg-max, ties to the lowest index like torch.max.
//   finalize  mutual test (+ min_cossim on the exact row maximum), ordered compaction          (k_match.hip, shared)
// A pair whose candidate list overflows (degenerate inputs: many identical descriptors) is redone by the exact f32 MFMA kernel;
// XFH_MATCH=f32 in the environment sends every pair there (A/B runs).
#include "kernels.hpp"

namespace xfh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BT_ROWS = 256;     // rows of D1 per workgroup (8 waves x 32)
constexpr int BT_COLS = 128;     // columns of D2 per LDS fill
constexpr int BT_DS = 72;        // LDS row stride in bf16 elements (144 bytes): the 16 lanes of a ds_read_b128 group hit 16 distinct 16-byte slots
constexpr float BF_C = 1.05f * 0.00390625f;      // c = 1.05 * 2^-8

__device__ inline int bpair_count(const int32_t* n, int idx, int cap) {
    if (!n) return cap;
    const int v = n[idx];
    return v < 0 ? 0 : (v > cap ? cap : v);
}
__device__ inline unsigned short f32_to_bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// 16 lanes per descriptor row (float4 each), 16 rows per pass, 256 rows per workgroup: bf16 copy, fp32 norm, per-pair maximum norm
// (one atomic per workgroup: a per-wave atomic on the pair's single word serialised 1024 of them -- 470 us).  grid (ceil(N/256), P, 2 sides)
__global__ __launch_bounds__(256) void mnn_prep_kernel(const float* __restrict__ d1, size_t ps1, const float* __restrict__ d2, size_t ps2,
                                                       const int32_t* __restrict__ n1p, const int32_t* __restrict__ n2p, int n_stride, int n_off2,
                                                       int N1, int N2, unsigned short* __restrict__ a16, unsigned short* __restrict__ b16,
                                                       float* __restrict__ na, float* __restrict__ nb, unsigned* __restrict__ nmax /* (2,P) */) {
    __shared__ float wmax[4];
    const int p = blockIdx.y, side = blockIdx.z, P = gridDim.y;
    const int N = side ? N2 : N1;
    const int n = side ? bpair_count(n2p, p * n_stride + n_off2, N2) : bpair_count(n1p, p * n_stride, N1);
    const int sub = threadIdx.x & 15;
    if (blockIdx.x * 256 >= n) return;
    const float* src = (side ? d2 + (size_t)p * ps2 : d1 + (size_t)p * ps1);
    float m = 0.f;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int row = blockIdx.x * 256 + it * 16 + (threadIdx.x >> 4);
        float s = 0.f;
        if (row < n) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)row * 64 + sub * 4);
            s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            ushort4 o;
            o.x = f32_to_bf16_rne(v.x); o.y = f32_to_bf16_rne(v.y); o.z = f32_to_bf16_rne(v.z); o.w = f32_to_bf16_rne(v.w);
            *reinterpret_cast<ushort4*>((side ? b16 : a16) + ((size_t)p * N + row) * 64 + sub * 4) = o;
        }
        s += __shfl_xor(s, 8, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 1, 64);
        const float nrm = sqrtf(s) * 1.000001f;             // (rounded up a hair: it only ever widens the window)
        if (row < n && sub == 0) (side ? nb : na)[(size_t)p * N + row] = nrm;
        if (row < n) m = fmaxf(m, nrm);
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)       // norms >= 0: the bit patterns order like the values
        atomicMax(&nmax[side * P + p], __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

// One 32x32 tile of S^ on four bf16 MFMAs.  a: this lane's A fragment (row l31, k = 16 kk + 8 half .. +7), bp: this lane's B row in LDS.
// The B fragments are handed back: the caller keeps them alive (an empty asm use) until its epilogue is over -- a VALU result written
// a few cycles after a K = 16 MFMA was issued can land in operand lanes the matrix core has not read yet, and hipcc reuses a dead
// fragment register for address arithmetic right behind the last MFMA (tools/check_mfma_war.py audits the generated code).
__device__ inline f32x16 bf16_tile(const bf16x8 (&a)[4], const unsigned short* bp, bf16x8 (&bfrag)[4]) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) bfrag[kk] = *reinterpret_cast<const bf16x8*>(bp + kk * 16);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk], bfrag[kk], acc, 0, 0, 0);
    return acc;
}
#define XFH_KEEP_FRAGS(f) asm volatile("" :: "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]))

// PASS 1: row / column maxima of S^.  PASS 2: candidates.  Same tiling as mnn_sim_kernel: a workgroup owns 256 rows, sweeps the columns.
template <int PASS>
__global__ __launch_bounds__(512) void mnn_bf16_kernel(const unsigned short* __restrict__ a16, size_t sa16, const unsigned short* __restrict__ b16, size_t sb16,
                                                       float unit_bound /* > 0: prepared unit-norm inputs, every |d| <= unit_bound (na / nb / nmax unused) */,
                                                       const int32_t* __restrict__ n1p, const int32_t* __restrict__ n2p, int n_stride, int n_off2,
                                                       int N1, int N2, int nrb, int P, const float* __restrict__ na, const float* __restrict__ nb,
                                                       const unsigned* __restrict__ nmax, float* __restrict__ rowmaxh, unsigned* __restrict__ colmaxh,
                                                       unsigned long long* __restrict__ cand, int cap, int* __restrict__ cnt) {
    __shared__ __attribute__((aligned(16))) unsigned short Dl[BT_COLS * BT_DS];
    __shared__ float colx[8][BT_COLS];                    // PASS 1: per-wave column maxima ; PASS 2: [0][] = column thresholds
    constexpr int LCAND = PASS == 2 ? 2048 : 1;           // PASS 2: the workgroup's candidates (one global atomic per workgroup: per-candidate
    __shared__ unsigned long long lcand[LCAND];           // atomics on the pair's single counter serialised ~10 k of them -- 1.2 ms)
    __shared__ int lcnt, lbase;
    if (PASS == 2 && threadIdx.x == 0) lcnt = 0;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int p, rb;
    if (!xcd_group_map(blockIdx.x, nrb, P, p, rb)) return;
    const int n1 = bpair_count(n1p, p * n_stride, N1);
    const int n2 = bpair_count(n2p, p * n_stride + n_off2, N2);
    const int row0 = rb * BT_ROWS;
    if (n1 <= 0 || n2 <= 0 || row0 >= n1) return;
    const unsigned short* A = a16 + (size_t)p * sa16;
    const unsigned short* Bm = b16 + (size_t)p * sb16;
    const int wrow0 = row0 + wave * 32;

    bf16x8 a[4];
    {
        const int row = min(wrow0 + l31, n1 - 1);          // rows >= n1: copies of the last valid row (never reported)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a[kk] = *reinterpret_cast<const bf16x8*>(A + (size_t)row * 64 + kk * 16 + half * 8);
    }
    float bv[16];                                          // PASS 1: running row maxima ; PASS 2: row thresholds
    if (PASS == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) bv[r] = -INFINITY;
    } else {
        const float e2 = 2.f * BF_C * (unit_bound > 0.f ? unit_bound : __uint_as_float(nmax[P + p]));          // 2 c max|d2|
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = min(wrow0 + (r & 3) + 8 * (r >> 2) + 4 * half, n1 - 1);
            bv[r] = rowmaxh[(size_t)p * N1 + row] - e2 * (unit_bound > 0.f ? unit_bound : na[(size_t)p * N1 + row]);
        }
    }
    const float e2c = PASS == 2 ? 2.f * BF_C * (unit_bound > 0.f ? unit_bound : __uint_as_float(nmax[p])) : 0.f;      // 2 c max|d1|

    for (int c0 = 0; c0 < n2; c0 += BT_COLS) {
        __syncthreads();
        {   // 128 columns x 64 bf16 = 1024 16-byte pieces, two per thread
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + i * 512;
                const int col = e >> 3, q = e & 7;
                const int gc = min(c0 + col, n2 - 1);
                const uint4 v = *reinterpret_cast<const uint4*>(Bm + (size_t)gc * 64 + q * 8);
                *reinterpret_cast<uint4*>(Dl + col * BT_DS + q * 8) = v;
            }
            if (PASS == 2 && tid < BT_COLS) {
                const int gc = min(c0 + tid, n2 - 1);
                colx[0][tid] = ord_float(colmaxh[(size_t)p * N2 + gc]) - e2c * (unit_bound > 0.f ? unit_bound : nb[(size_t)p * N2 + gc]);
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int ct = 0; ct < BT_COLS / 32; ++ct) {
            const int cbase = c0 + ct * 32;
            if (cbase >= n2) break;
            bf16x8 bfrag[4];
            const f32x16 acc = bf16_tile(a, Dl + (ct * 32 + l31) * BT_DS + half * 8, bfrag);
            // D[i=row][j=col]: this lane holds column cbase+l31, rows (r&3)+8*(r>>2)+4*half
            if (PASS == 1) {
                float cm = acc[0];
#pragma unroll
                for (int r = 0; r < 16; ++r) { bv[r] = fmaxf(bv[r], acc[r]); cm = fmaxf(cm, acc[r]); }
                cm = fmaxf(cm, xhalf(cm));
                if (half == 0) colx[wave][ct * 32 + l31] = cm;
            } else {
                // hit masks straight out of the compares (v_cmp -> SGPR pair, OR-ed on the scalar unit): two VALU ops per element
                const float cthr = colx[0][ct * 32 + l31];
                const int col = cbase + l31;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {             // rows in groups of four: ~0.6 candidates per tile, most groups skip on a scalar branch
                    unsigned long long mk[4], any = 0ull;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = 4 * g4 + q;
                        mk[q] = __ballot(acc[r] >= bv[r]) | __ballot(acc[r] >= cthr);
                        any |= mk[q];
                    }
                    if (any) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (((mk[q] >> lane) & 1ull) && col < n2) {           // padding columns are copies: not candidates
                                const int row = wrow0 + q + 8 * g4 + 4 * half;    // (r&3) + 8*(r>>2) + 4*half with r = 4*g4 + q
                                if (row < n1) {
                                    const int idx = atomicAdd(&lcnt, 1);
                                    if (idx < LCAND) lcand[idx] = ((unsigned long long)(unsigned)row << 32) | (unsigned)col;
                                }
                            }
                    }
                }
            }
            XFH_KEEP_FRAGS(bfrag);
        }
        if (PASS == 1) {
            __syncthreads();
            if (tid < BT_COLS) {
                const int col = c0 + tid;
                if (col < n2) {
                    float k = colx[0][tid];
#pragma unroll
                    for (int w = 1; w < 8; ++w) k = fmaxf(k, colx[w][tid]);
                    atomicMax(&colmaxh[(size_t)p * N2 + col], float_ord(k));
                }
            }
        }
    }
    if (PASS == 2) {
        __syncthreads();
        const int mine = lcnt;
        // a local overflow forces the pair's overflow (count beyond cap): the exact kernel redoes it
        if (tid == 0) lbase = atomicAdd(&cnt[p], mine > LCAND ? cap + 1 : mine);
        __syncthreads();
        const int base = lbase;
        for (int i = tid; i < min(mine, LCAND); i += 512)
            if (base + i < cap) cand[(size_t)p * cap + base + i] = lcand[i];
    }
    if (PASS == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = bv[r];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
            const int row = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (l31 == 0 && row < n1) rowmaxh[(size_t)p * N1 + row] = v;
        }
    }
}

// exact fp32 dot product of every candidate, 4 lanes per candidate (16 channels each); grid (wgs_per_pair, P)
__global__ __launch_bounds__(256) void mnn_exact_kernel(const float* __restrict__ d1, size_t ps1, const float* __restrict__ d2, size_t ps2,
                                                        const unsigned long long* __restrict__ cand, int cap, const int* __restrict__ cnt,
                                                        int N1, int N2, unsigned long long* __restrict__ rowkey, unsigned long long* __restrict__ colkey) {
    const int p = blockIdx.y;
    const int total = cnt[p];
    if (total > cap) return;                                // overflow: the exact MFMA kernel redoes this pair
    const int q = threadIdx.x & 3;
    const float* A = d1 + (size_t)p * ps1;
    const float* Bm = d2 + (size_t)p * ps2;
    for (int c = blockIdx.x * 64 + (threadIdx.x >> 2); c < total; c += gridDim.x * 64) {
        const unsigned long long e = cand[(size_t)p * cap + c];
        const unsigned i = (unsigned)(e >> 32), j = (unsigned)e;
        const float4* ap = reinterpret_cast<const float4*>(A + (size_t)i * 64 + q * 16);
        const float4* bp = reinterpret_cast<const float4*>(Bm + (size_t)j * 64 + q * 16);
        float4 x[4], y[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { x[t] = ap[t]; y[t] = bp[t]; }
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            s = fmaf(x[t].x, y[t].x, s); s = fmaf(x[t].y, y[t].y, s);
            s = fmaf(x[t].z, y[t].z, s); s = fmaf(x[t].w, y[t].w, s);
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (q == 0) {
            const unsigned long long hi = (unsigned long long)float_ord(s) << 32;
            atomicMax(&rowkey[(size_t)p * N1 + i], hi | (0xffffffffu - j));
            atomicMax(&colkey[(size_t)p * N2 + j], hi | (0xffffffffu - i));
        }
    }
}

// d1_16 / d2_16 (optional, both or neither): bf16 RNE copies the caller already holds (xfh_detect_sparse's desc_bf16), laid out like d1 / d2
// (same pair strides in elements), rows L2-normalised: |d| <= 1.00001.  They replace the prep pass.
void launch_match_bf16(const MatchWs& ws, const float* d1, size_t ps1, const float* d2, size_t ps2, const uint16_t* d1_16, const uint16_t* d2_16,
                       const int32_t* n1, const int32_t* n2, int n_stride, int n_off2, int P, int N1, int N2, hipStream_t st) {
    const int nrb = ceil_div(N1, BT_ROWS);
    const bool prepared = d1_16 && d2_16;
    const unsigned short* a16 = prepared ? d1_16 : ws.a16;
    const unsigned short* b16 = prepared ? d2_16 : ws.b16;
    const size_t sa = prepared ? ps1 : (size_t)N1 * 64, sb = prepared ? ps2 : (size_t)N2 * 64;
    const float ub = prepared ? 1.00001f : 0.f;
    if (!prepared)
        mnn_prep_kernel<<<dim3(ceil_div(N1 > N2 ? N1 : N2, 256), P, 2), 256, 0, st>>>(d1, ps1, d2, ps2, n1, n2, n_stride, n_off2, N1, N2, ws.a16, ws.b16,
                                                                                    ws.na, ws.nb, ws.nmax);
    mnn_bf16_kernel<1><<<xcd_grid_size(nrb, P), 512, 0, st>>>(a16, sa, b16, sb, ub, n1, n2, n_stride, n_off2, N1, N2, nrb, P, ws.na, ws.nb, ws.nmax,
                                                             ws.rowmaxh, ws.colmaxh, ws.cand, ws.cand_cap, ws.cnt);
    mnn_bf16_kernel<2><<<xcd_grid_size(nrb, P), 512, 0, st>>>(a16, sa, b16, sb, ub, n1, n2, n_stride, n_off2, N1, N2, nrb, P, ws.na, ws.nb, ws.nmax,
                                                             ws.rowmaxh, ws.colmaxh, ws.cand, ws.cand_cap, ws.cnt);
    mnn_exact_kernel<<<dim3(64, P), 256, 0, st>>>(d1, ps1, d2, ps2, ws.cand, ws.cand_cap, ws.cnt, N1, N2, ws.rowkey, ws.colkey);
}

}  // namespace xfh
