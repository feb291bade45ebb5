// Mutual-nearest-neighbour matching, filter-and-refine form (the shipped path of xfh_match_mnn).
//   XFeat.match        modules/xfeat.py:327-348     XFeat.batch_match  modules/xfeat.py:265-290
//
// The exact kernel (k_match.hip) spends 32 f32 MFMAs (2048 pipe cycles) on every 32x32 tile of S = D1.D2^T to learn two things per
// row / column: the arg-max.  Here the matrix cores are a FILTER and every decision is still taken on exact fp32 dot products:
//   prep      (only when the caller has no fp16 copies) norms, the pair's maximum norms, then D1, D2 -> fp16 copies of s.D
//             (round-to-nearest-even; s = a power of two per pair and side that puts the largest row norm in [128, 256])
//   sweep     ONE pass over S^ = D1^.D2^T on v_mfma_f32_32x32x16_f16 (exact fp16 products, fp32 accumulation), every 32x32 tile in
//             BOTH orientations -- mfma(a, b) leaves a column of the tile in each lane, mfma(b, a) a row: the two fragments have the
//             same register layout, so the transposed tile costs four more MFMAs and no loads -- which turns both reductions into
//             in-lane v_max3 trees (8 ops for 16 values):
//               C[rb][j] = max of S^ over the 32 rows of row block rb, column j       R[cb][i] = max over the 32 columns of block cb, row i
//             plus the row maxima (running, one op per tile) and the column maxima (LDS across the waves, one atomic per column and
//             workgroup).  8 MFMAs (256 pipe cycles) and ~28 VALU ops per tile: the matrix pipe is the bound.
//   refine    for every row i: the column blocks with R[cb][i] >= rowmax^_i - 2 E_i are the only ones that can hold the exact arg-max
//             of row i (or an exact tie with it); all 32 of their fp32 dot products are computed (fixed summation order) and folded
//             into the row key (ord(S) << 32 | ~j) with a 64-bit atomic max -- ties to the lowest index like torch.max.  Columns
//             likewise from C.  ~1.05 blocks per row / column on descriptor data; identical descriptors flag every block: slower,
//             still exact, no capacity to overflow.
//   finalize  mutual test (+ min_cossim on the exact row maximum), ordered compaction          (k_match.hip, shared)
//
// The window.  a, b: two rows (fp32), a^ = a + alpha, b^ = b + beta what the matrix core multiplies (in units of a, b).  fp16 has an
// 11-bit significand: round-to-nearest-even gives |err| <= u |v| with u = 2^-11 for normal results; results below 2^-14 (subnormal,
// or flushed to zero by the matrix core -- either way) are off by at most 2^-14.  With the scale s >= 2^7 / maxnorm that is an
// absolute tau <= 2^-21 maxnorm per component, so |alpha_k| <= u |a_k| + tau_a, |beta_k| <= u |b_k| + tau_b.  Exactly,
//     a.b - a^.b^ = sum_k alpha_k b_k + a^_k beta_k ,
// hence with Cauchy-Schwarz and |v|_1 <= 8 |v|_2 in 64 dimensions
//     |S - S^| <= (2u + u^2) |a||b| + 9 (tau_a |b| + tau_b |a|) <= (2^-10 + 2^-22) |a||b| + 9 * 2^-20 maxnorm_a maxnorm_b .
// The fp32 accumulation inside the MFMA adds at most 64 * 2^-23 (1+u)^2 |a||b| (truncating adds, any order) and the refine's own fp32
// dot product at most 18 * 2^-24 |a||b|.  All of it is below
//     E(a, b) = c |a||b| + kappa maxnorm_a maxnorm_b ,   c = 1.03 * 2^-10 ,  kappa = 1e-5        (2 % head-room on c).
// For the refine's arg-max j* of row i and the filter's arg-max j^:  S^_ij* >= S_ij* - E >= S_ij^ - E >= S^_ij^ - 2E = rowmax^_i - 2E
// with E = E_i = c |a_i| max_j|b_j| + kappa maxnorm_a maxnorm_b, and the block maximum R[cb(j*)][i] >= S^_ij*: the block of j* is
// flagged -- the same for every exact tie with j* and, with the roles swapped, for columns.  (Round 2 shipped a bf16 filter whose window
// assumed u = 2^-9; bf16 RNE has u = 2^-8, so that window was a factor 2 short of a proof.  fp16 on unit-norm rows is 8x tighter than
// bf16 and the window above is derived, and tested on rounding-aligned adversarial rows: tests/test_gpu_parity.py.)
#include "kernels.hpp"

namespace xfh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int FT_ROWS = 256;     // rows of D1 per workgroup (8 waves x 32)
constexpr int FT_COLS = 128;     // columns of D2 per LDS fill
constexpr int FT_DS = 72;        // LDS row stride in fp16 elements (144 bytes): the 16 lanes of a ds_read_b128 group hit 16 distinct 16-byte slots
constexpr float F16_C = 1.03f * 0.0009765625f;       // c = 1.03 * 2^-10
constexpr float F16_KAPPA = 1.0e-5f;
constexpr float F16_UNIT_SCALE = 256.f;              // scale of caller-provided copies of unit-norm rows (xfh_detect_sparse's desc_f16)

__device__ inline int fpair_count(const int32_t* n, int idx, int cap) {
    if (!n) return cap;
    const int v = n[idx];
    return v < 0 ? 0 : (v > cap ? cap : v);
}
// power of two s with 128 <= s * maxnorm <= 256 (maxnorm > 0), 1 for an all-zero set
__device__ inline float f16_scale(float maxnorm) {
    if (!(maxnorm > 0.f)) return 1.f;
    int e;
    (void)frexpf(maxnorm, &e);                       // maxnorm = m 2^e, m in [0.5, 1)
    return ldexpf(1.f, 8 - e);
}

// 16 lanes per descriptor row (float4 each), 16 rows per pass, 256 rows per workgroup.  grid (ceil(N/256), P, 2 sides)
// CVT = false: fp32 norms (rounded up a hair: that only ever widens the window) and the pair's maximum norm, one atomic per workgroup
// CVT = true : fp16 copies of s * row
template <bool CVT>
__global__ __launch_bounds__(256) void mnn_prep_kernel(const float* __restrict__ d1, size_t ps1, const float* __restrict__ d2, size_t ps2,
                                                       const int32_t* __restrict__ n1p, const int32_t* __restrict__ n2p, int n_stride, int n_off2,
                                                       int N1, int N2, _Float16* __restrict__ a16, _Float16* __restrict__ b16,
                                                       float* __restrict__ na, float* __restrict__ nb, unsigned* __restrict__ nmax /* (2,P) */) {
    __shared__ float wmax[4];
    const int p = blockIdx.y, side = blockIdx.z, P = gridDim.y;
    const int N = side ? N2 : N1;
    const int n = side ? fpair_count(n2p, p * n_stride + n_off2, N2) : fpair_count(n1p, p * n_stride, N1);
    const int sub = threadIdx.x & 15;
    if (blockIdx.x * 256 >= n) return;
    const float* src = (side ? d2 + (size_t)p * ps2 : d1 + (size_t)p * ps1);
    const float sc = CVT ? f16_scale(__uint_as_float(nmax[side * P + p])) : 0.f;
    float m = 0.f;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int row = blockIdx.x * 256 + it * 16 + (threadIdx.x >> 4);
        float s = 0.f;
        if (row < n) {
            const float4 v = *reinterpret_cast<const float4*>(src + (size_t)row * 64 + sub * 4);
            if (CVT) {
                f16x4 o;
                o[0] = (_Float16)(v.x * sc); o[1] = (_Float16)(v.y * sc); o[2] = (_Float16)(v.z * sc); o[3] = (_Float16)(v.w * sc);
                *reinterpret_cast<f16x4*>((side ? b16 : a16) + ((size_t)p * N + row) * 64 + sub * 4) = o;
            } else {
                s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
        }
        if (!CVT) {
            s += __shfl_xor(s, 8, 64);
            s += __shfl_xor(s, 4, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 1, 64);
            const float nrm = sqrtf(s) * 1.000001f;
            if (row < n && sub == 0) (side ? nb : na)[(size_t)p * N + row] = nrm;
            if (row < n) m = fmaxf(m, nrm);
        }
    }
    if (!CVT) {
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0)       // norms >= 0: the bit patterns order like the values
            atomicMax(&nmax[side * P + p], __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
    }
}

// maximum of 16 accumulator values as a v_max3_f32 tree (8 ops)
__device__ inline float max16(const f32x16& v) {
    const float a = fmaxf(fmaxf(v[0], v[1]), v[2]);
    const float b = fmaxf(fmaxf(v[3], v[4]), v[5]);
    const float c = fmaxf(fmaxf(v[6], v[7]), v[8]);
    const float d = fmaxf(fmaxf(v[9], v[10]), v[11]);
    const float e = fmaxf(fmaxf(v[12], v[13]), v[14]);
    const float f = fmaxf(fmaxf(a, b), c);
    const float g = fmaxf(fmaxf(d, e), v[15]);
    return fmaxf(f, g);
}

// E in the units of the scaled product S^' = (sa sb) S^ :  2 E' = 2 sa sb (c |x| maxnorm_y + kappa maxnorm_x maxnorm_y)
struct PairWindow { float two_c, two_k; };
__device__ inline PairWindow pair_window(float unit_bound, const unsigned* __restrict__ nmax, int P, int p, bool row_side) {
    PairWindow w;
    if (unit_bound > 0.f) {          // caller-provided copies of rows with |row| <= unit_bound, scale 256 on both sides
        const float ss = F16_UNIT_SCALE * F16_UNIT_SCALE;
        w.two_c = 2.f * F16_C * ss * unit_bound;                    // times |x| (= unit_bound)
        w.two_k = 2.f * F16_KAPPA * ss * unit_bound * unit_bound;
    } else {
        const float ma = __uint_as_float(nmax[p]), mb = __uint_as_float(nmax[P + p]);
        const float ss = f16_scale(ma) * f16_scale(mb);
        w.two_c = 2.f * F16_C * ss * (row_side ? mb : ma);          // times |x|
        w.two_k = 2.f * F16_KAPPA * ss * ma * mb;
    }
    return w;
}

// The B fragments of a tile are kept alive (an empty asm use) until the tile's epilogue is over: a VALU result written a few cycles
// after a K = 16 MFMA was issued can land in operand lanes the matrix core has not read yet, and hipcc reuses a dead fragment register
// for address arithmetic right behind the last MFMA (tools/check_mfma_war.py audits the generated code).
#define XFH_KEEP_FRAGS(f) asm volatile("" :: "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]))

// A workgroup owns 256 rows of D1 and sweeps the columns of D2 (staged 128 at a time through LDS).
//   thr_row (P,N1)      : rowmax^_i - 2 E_i          (scaled units; the workgroup sees whole rows)
//   colmaxh (P,N2) u32  : ord(column maximum), 32-bit atomic max across the row-block workgroups (zeroed by the caller)
//   R (P, ceil(N2/32), N1), C (P, ceil(N1/32), N2) : block maxima
__global__ __launch_bounds__(512) void mnn_f16_sweep_kernel(const _Float16* __restrict__ a16, size_t sa16, const _Float16* __restrict__ b16, size_t sb16,
                                                            float unit_bound /* > 0: caller-provided copies of unit-norm rows (na / nb / nmax unused) */,
                                                            const int32_t* __restrict__ n1p, const int32_t* __restrict__ n2p, int n_stride, int n_off2,
                                                            int N1, int N2, int nrb, int P, const float* __restrict__ na, const unsigned* __restrict__ nmax,
                                                            float* __restrict__ thr_row, unsigned* __restrict__ colmaxh,
                                                            float* __restrict__ R, float* __restrict__ C) {
    __shared__ __attribute__((aligned(16))) _Float16 Dl[FT_COLS * FT_DS];
    __shared__ float colx[8][FT_COLS];                    // per-wave column maxima of the current 128 columns
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int p, rb;
    if (!xcd_group_map(blockIdx.x, nrb, P, p, rb)) return;
    const int n1 = fpair_count(n1p, p * n_stride, N1);
    const int n2 = fpair_count(n2p, p * n_stride + n_off2, N2);
    const int row0 = rb * FT_ROWS;
    if (n1 <= 0 || n2 <= 0 || row0 >= n1) return;
    const _Float16* A = a16 + (size_t)p * sa16;
    const _Float16* Bm = b16 + (size_t)p * sb16;
    const int wrow0 = row0 + wave * 32;
    const bool wave_live = wrow0 < n1;                    // waves past the last row still stage and meet the barriers
    const int myrow = wrow0 + l31;
    const int ncb32 = ceil_div(N2, 32), nrb32 = ceil_div(N1, 32);
    float* Cw = C + ((size_t)p * nrb32 + (wrow0 >> 5)) * N2;
    float* Rw = R + (size_t)p * ncb32 * N1 + myrow;

    f16x8 a[4];
    {
        const int row = min(myrow, n1 - 1);               // rows >= n1: copies of the last valid row (never reported)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a[kk] = *reinterpret_cast<const f16x8*>(A + (size_t)row * 64 + kk * 16 + half * 8);
    }
    float rowrun = -INFINITY;

    for (int c0 = 0; c0 < n2; c0 += FT_COLS) {
        __syncthreads();
        {   // 128 columns x 64 fp16 = 1024 16-byte pieces, two per thread
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = tid + i * 512;
                const int col = e >> 3, q = e & 7;
                const int gc = min(c0 + col, n2 - 1);      // columns >= n2: copies of the last valid column
                const uint4 v = *reinterpret_cast<const uint4*>(Bm + (size_t)gc * 64 + q * 8);
                *reinterpret_cast<uint4*>(Dl + col * FT_DS + q * 8) = v;
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int ct = 0; ct < FT_COLS / 32; ++ct) {
            const int cbase = c0 + ct * 32;
            if (cbase >= n2) break;
            f16x8 bfrag[4];
            const _Float16* bp = Dl + (ct * 32 + l31) * FT_DS + half * 8;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) bfrag[kk] = *reinterpret_cast<const f16x8*>(bp + kk * 16);
            f32x16 acc, accT;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accT[r] = 0.f; }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], bfrag[kk], acc, 0, 0, 0);      // lane: column cbase + l31, 16 of the wave's rows
                accT = __builtin_amdgcn_mfma_f32_32x32x16_f16(bfrag[kk], a[kk], accT, 0, 0, 0);    // lane: row wrow0 + l31, 16 of the tile's columns
            }
            float cm = max16(acc), rm = max16(accT);
            cm = fmaxf(cm, xhalf(cm));
            rm = fmaxf(rm, xhalf(rm));
            rowrun = fmaxf(rowrun, rm);
            if (half == 0) {
                colx[wave][ct * 32 + l31] = cm;
                if (wave_live && cbase + l31 < n2) Cw[cbase + l31] = cm;
            } else if (myrow < n1) {
                Rw[(size_t)(cbase >> 5) * N1] = rm;
            }
            XFH_KEEP_FRAGS(bfrag);
        }
        __syncthreads();
        if (tid < FT_COLS) {
            const int col = c0 + tid;
            if (col < n2) {
                float k = colx[0][tid];
#pragma unroll
                for (int w = 1; w < 8; ++w) k = fmaxf(k, colx[w][tid]);
                atomicMax(&colmaxh[(size_t)p * N2 + col], float_ord(k));
            }
        }
    }
    if (half == 0 && myrow < n1) {
        const PairWindow w = pair_window(unit_bound, nmax, P, p, true);
        const float nx = unit_bound > 0.f ? unit_bound : na[(size_t)p * N1 + myrow];
        thr_row[(size_t)p * N1 + myrow] = rowrun - fmaf(w.two_c, nx, w.two_k);
    }
}

// Exact refine.  Workgroup (pair, side, yb): the 32 rows yb*32.. of the OTHER set (Y) sit in LDS; every x of the own set whose block
// maximum reaches its threshold gets the 32 exact dot products with them, folded into key[x] = max (ord(S) << 32 | ~y).
//   side 0: x = rows of D1 (keys: row arg-max), Y = D2, block maxima R, thresholds thr_row
//   side 1: x = rows of D2 (keys: column arg-max), Y = D1, block maxima C, thresholds from colmaxh (complete only now)
// The dot product is the one the round-2 refine used, bit for bit: four chains of 16 channels, (s0 + s1) + (s2 + s3).
constexpr int RF_LIST = 2048;
constexpr int RF_YS = 68;          // LDS row stride in floats (272 bytes = 17 x 16: the lanes of a ds_read_b128 group hit distinct slots)
__global__ __launch_bounds__(256) void mnn_f16_refine_kernel(const float* __restrict__ d1, size_t ps1, const float* __restrict__ d2, size_t ps2,
                                                             float unit_bound, const int32_t* __restrict__ n1p, const int32_t* __restrict__ n2p,
                                                             int n_stride, int n_off2, int N1, int N2, int nyb_max, int P,
                                                             const float* __restrict__ na, const float* __restrict__ nb, const unsigned* __restrict__ nmax,
                                                             const float* __restrict__ thr_row, const unsigned* __restrict__ colmaxh,
                                                             const float* __restrict__ R, const float* __restrict__ C,
                                                             unsigned long long* __restrict__ rowkey, unsigned long long* __restrict__ colkey) {
    __shared__ __attribute__((aligned(16))) float Ys[32 * RF_YS];
    __shared__ unsigned short list[RF_LIST];
    __shared__ int lcnt;
    int p, item;
    if (!xcd_group_map(blockIdx.x, 2 * nyb_max, P, p, item)) return;
    const int side = item / nyb_max, yb = item - side * nyb_max;
    const int n1 = fpair_count(n1p, p * n_stride, N1);
    const int n2 = fpair_count(n2p, p * n_stride + n_off2, N2);
    if (n1 <= 0 || n2 <= 0) return;
    const int nX = side ? n2 : n1, nY = side ? n1 : n2, NX = side ? N2 : N1;
    if (yb * 32 >= nY) return;
    const float* X = side ? d2 + (size_t)p * ps2 : d1 + (size_t)p * ps1;
    const float* Y = side ? d1 + (size_t)p * ps1 : d2 + (size_t)p * ps2;
    const float* M = (side ? C + (size_t)p * ceil_div(N1, 32) * N2 : R + (size_t)p * ceil_div(N2, 32) * N1) + (size_t)yb * NX;
    unsigned long long* key = side ? colkey + (size_t)p * N2 : rowkey + (size_t)p * N1;
    const int tid = threadIdx.x;
    {   // 32 rows x 64 floats = 512 float4, two per thread; rows past nY: zeros (their keys are never written)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * 256;
            const int r = e >> 4, q = e & 15;
            const int gy = yb * 32 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy < nY) v = *reinterpret_cast<const float4*>(Y + (size_t)gy * 64 + q * 4);
            *reinterpret_cast<float4*>(Ys + r * RF_YS + q * 4) = v;
        }
    }
    const PairWindow w = pair_window(unit_bound, nmax, P, p, side == 0);
    const float* nx = side ? nb + (size_t)p * N2 : na + (size_t)p * N1;
    const int grp = tid >> 5, l31 = tid & 31;
    const int gy = yb * 32 + l31;
    for (int x0 = 0; x0 < nX; x0 += RF_LIST) {
        if (tid == 0) lcnt = 0;
        __syncthreads();                                   // (also covers the Ys fill the first time round)
        {   // scan: RF_LIST block maxima against their thresholds, all loads of a thread in flight together
            constexpr int NL = RF_LIST / 256;
            float m[NL], t[NL];
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const int x = x0 + tid + k * 256;
                m[k] = -INFINITY; t[k] = INFINITY;
                if (x < nX) {
                    m[k] = M[x];
                    if (side == 0) t[k] = thr_row[(size_t)p * N1 + x];
                    else t[k] = ord_float(colmaxh[(size_t)p * N2 + x]) - fmaf(w.two_c, unit_bound > 0.f ? unit_bound : nx[x], w.two_k);
                }
            }
#pragma unroll
            for (int k = 0; k < NL; ++k)
                if (m[k] >= t[k]) list[atomicAdd(&lcnt, 1)] = (unsigned short)(tid + k * 256);
        }
        __syncthreads();
        const int cnt = lcnt;
        for (int e = grp; e < cnt; e += 8) {               // a half-wave per flagged x: lane = row of the Y block
            const int x = x0 + list[e];
            const float4* xp = reinterpret_cast<const float4*>(X + (size_t)x * 64);      // one address per half-wave: a broadcast load
            const float4* yp = reinterpret_cast<const float4*>(Ys + l31 * RF_YS);
            float s[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 xv[4], yv[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { xv[t] = xp[q * 4 + t]; yv[t] = yp[q * 4 + t]; }
                float c = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    c = fmaf(xv[t].x, yv[t].x, c); c = fmaf(xv[t].y, yv[t].y, c);
                    c = fmaf(xv[t].z, yv[t].z, c); c = fmaf(xv[t].w, yv[t].w, c);
                }
                s[q] = c;
            }
            const float dot = (s[0] + s[1]) + (s[2] + s[3]);
            unsigned long long k = gy < nY ? ((unsigned long long)float_ord(dot) << 32) | (0xffffffffu - (unsigned)gy) : 0ull;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) k = u64_max(k, shfl_xor_u64(k, o));
            if (l31 == 0) atomicMax(&key[x], k);
        }
        __syncthreads();
    }
}

// d1_16 / d2_16 (optional, both or neither): fp16 copies the caller already holds (xfh_detect_sparse's desc_f16 = RNE(256 * row)), laid out
// like d1 / d2 (same pair strides in elements), rows L2-normalised: |row| <= 1.00001.  They replace the prep passes.
void launch_match_f16(const MatchWs& ws, const float* d1, size_t ps1, const float* d2, size_t ps2, const uint16_t* d1_16, const uint16_t* d2_16,
                      const int32_t* n1, const int32_t* n2, int n_stride, int n_off2, int P, int N1, int N2, hipStream_t st) {
    const int nrb = ceil_div(N1, FT_ROWS);
    const bool prepared = d1_16 && d2_16;
    const _Float16* a16 = prepared ? reinterpret_cast<const _Float16*>(d1_16) : ws.a16;
    const _Float16* b16 = prepared ? reinterpret_cast<const _Float16*>(d2_16) : ws.b16;
    const size_t sa = prepared ? ps1 : (size_t)N1 * 64, sb = prepared ? ps2 : (size_t)N2 * 64;
    const float ub = prepared ? 1.00001f : 0.f;
    if (!prepared) {
        const dim3 g(ceil_div(N1 > N2 ? N1 : N2, 256), P, 2);
        mnn_prep_kernel<false><<<g, 256, 0, st>>>(d1, ps1, d2, ps2, n1, n2, n_stride, n_off2, N1, N2, ws.a16, ws.b16, ws.na, ws.nb, ws.nmax);
        mnn_prep_kernel<true><<<g, 256, 0, st>>>(d1, ps1, d2, ps2, n1, n2, n_stride, n_off2, N1, N2, ws.a16, ws.b16, ws.na, ws.nb, ws.nmax);
    }
    mnn_f16_sweep_kernel<<<xcd_grid_size(nrb, P), 512, 0, st>>>(a16, sa, b16, sb, ub, n1, n2, n_stride, n_off2, N1, N2, nrb, P, ws.na, ws.nmax,
                                                               ws.thr_row, ws.colmaxh, ws.R, ws.C);
    const int nyb = ceil_div(N1 > N2 ? N1 : N2, 32);
    mnn_f16_refine_kernel<<<xcd_grid_size(2 * nyb, P), 256, 0, st>>>(d1, ps1, d2, ps2, ub, n1, n2, n_stride, n_off2, N1, N2, nyb, P, ws.na, ws.nb, ws.nmax,
                                                                    ws.thr_row, ws.colmaxh, ws.R, ws.C, ws.rowkey, ws.colkey);
}

}  // namespace xfh
