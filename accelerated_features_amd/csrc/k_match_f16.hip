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
//             plus the column maxima (running in registers; the workgroup sees whole columns) and the row maxima (one 32-bit atomic
//             per row and workgroup).  8 MFMAs (256 pipe cycles) and ~28 VALU ops per tile: the matrix pipe is the bound.
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
#include "../../include/xfeat_hip.h"
#include "kernels.hpp"
#include <type_traits>

namespace xfh {

void prof_begin(Profiler* p, int which, hipStream_t st);
void prof_end(Profiler* p, int which, hipStream_t st, double flops, double bytes);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));


constexpr int FT_COLS = 256;     // columns of D2 per LDS fill
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

// Block maxima and thresholds travel as fp16 (round 5: they were fp32): the sweep writes, and the refine's scan reads, half the bytes -- 67 instead of 134 MB per
// 64-frame step.  The comparison stays CONSERVATIVE: a block maximum is rounded UP, a threshold DOWN, so every block with R >= thr in fp32 still has R16 >= thr16 (the
// flagged set only grows -- by the blocks within ~2^-9 |value| of the threshold: the refine evaluates a few more blocks exactly, decisions are taken on fp32 dot products
// as before).  Scaled products reach 2^16 (unit rows at scale 256 on both sides), fp16 ends at 65504: the stored value is S^' / 4 (exact).
//   up(x):   y = x + 2^-10 |x| + 2^-24, then round-to-nearest-even to fp16.  RNE moves a normal y by at most half an ulp <= 2^-11 |y|, a subnormal one by at most 2^-25:
//            fp16(y) >= y - max(2^-11 |y|, 2^-25) >= x.        down(x) = -up(-x).
constexpr float F16_STORE_SCALE = 0.25f;
__device__ inline _Float16 f16_up(float x) { return (_Float16)(__builtin_fmaf(__builtin_fabsf(x), 0.0009765625f, x) + 5.9604644775390625e-8f); }
__device__ inline _Float16 f16_down(float x) { return (_Float16)(__builtin_fmaf(__builtin_fabsf(x), -0.0009765625f, x) - 5.9604644775390625e-8f); }

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

// maximum of a value and the other half-wave's (lane ^ 32): v_permlane32_swap of two copies leaves {lo, lo} in one register and {hi, hi} in the other -- their maximum is
// the answer in both halves (no select)
__device__ inline float xhalf_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
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
    // wave-uniform by construction: keep them in scalar registers
    w.two_c = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(w.two_c)));
    w.two_k = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(w.two_k)));
    return w;
}

// The B fragments of a tile are kept alive (an empty asm use) until the tile's epilogue is over: a VALU result written a few cycles
// after a K = 16 MFMA was issued can land in operand lanes the matrix core has not read yet, and hipcc reuses a dead fragment register
// for address arithmetic right behind the last MFMA (tools/check_mfma_war.py audits the generated code).
#define XFH_KEEP_FRAGS(f) asm volatile("" :: "v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]))
#define XFH_WAVE_SYNC() __builtin_amdgcn_wave_barrier()      // (LDS operations of one wave execute in order; this keeps the compiler from reordering them)

// Sweep.  A workgroup keeps 256 columns (rows of D2) in LDS for its whole life and streams the 32-row blocks of D1 past them: no barrier
// inside the loop -- the waves take row blocks from a shared counter, so a wave the scheduler favours simply does more of them.  Per block
// the A fragments come straight from global memory into registers, one block ahead.  Per 32x32 tile the two orientations run one after the
// other on 16 accumulator registers each, and the VALU epilogue of one hides in the issue gaps of the other's MFMAs (a K = 16 MFMA holds
// the pipe for 32 cycles = room for ~5 other instructions of the same wave; with the epilogues AFTER all eight MFMAs the kernel measured
// MFMA time + VALU time, the pipe 53 % busy: profiles/r03_*):
//     A:  acc  = a . b            (lane = column)                        |  B:  accT = b . a  (lane = row)  ||  column epilogue of acc
//     then the fragments of the next tile leave LDS                       ||  row epilogue of accT
//   colmaxh (P,N2) u32  : ord(column maximum), 32-bit atomic max across the row shares (zeroed by the caller)
//   rowmaxh (P,N1) u32  : ord(row maximum), 32-bit atomic max across the column-chunk workgroups (zeroed by the caller)
//   R (P, ceil(N2/32), N1), C (P, ceil(N1/32), N2) : block maxima, fp16, a quarter of the scaled product rounded UP (f16_up)
constexpr int FT_TILES = FT_COLS / 32;
__global__ __launch_bounds__(512) void mnn_f16_sweep_kernel(const _Float16* __restrict__ a16, size_t sa16, const _Float16* __restrict__ b16, size_t sb16,
                                                            const int32_t* __restrict__ n1p, const int32_t* __restrict__ n2p, int n_stride, int n_off2,
                                                            int N1, int N2, int ncc, int nsplit, int P,
                                                            unsigned* __restrict__ colmaxh, unsigned* __restrict__ rowmaxh,
                                                            _Float16* __restrict__ R, _Float16* __restrict__ C
#if XFH_CODE_SHIFT > 0      // torture builds only (build.py --shift N): the production kernel is, byte for byte, the one the round-4 proof soaks ran
                                                            , int cold
#endif
                                                            ) {
#if XFH_CODE_SHIFT > 0
    kernel_entry_hooks(cold);      // debug: code-position shift / cold instruction cache (common.hpp)
#endif
    __shared__ __attribute__((aligned(16))) _Float16 Dl[FT_COLS * FT_DS];      // the columns
    __shared__ float colx[8][FT_COLS];                                          // per wave: running column maxima (ds_max_f32, no return)
    __shared__ int next_block;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int p, item;
    if (!xcd_group_map(blockIdx.x, ncc * nsplit, P, p, item)) return;
    const int cc = item / nsplit, rs = item - cc * nsplit;     // column chunk, and which share of the row blocks (few pairs: more workgroups)
    const int n1 = fpair_count(n1p, p * n_stride, N1);
    const int n2 = fpair_count(n2p, p * n_stride + n_off2, N2);
    const int c0 = cc * FT_COLS;
    if (n1 <= 0 || n2 <= 0 || c0 >= n2) return;
    const _Float16* A = a16 + (size_t)p * sa16;
    const _Float16* Bm = b16 + (size_t)p * sb16;
    const int ncb32 = ceil_div(N2, 32), nrb32 = ceil_div(N1, 32);
    const int ntile = min(FT_TILES, ceil_div(n2 - c0, 32));
    const int bps = ceil_div(ceil_div(n1, 32), nsplit);     // row blocks per share
    const int blk_lo = rs * bps, nblock = min(ceil_div(n1, 32), blk_lo + bps);
    if (blk_lo >= nblock) return;
    if (tid == 0) next_block = blk_lo + 8;                 // the first eight blocks: one per wave to start with
    {   // 256 columns x 64 fp16 = 2048 16-byte pieces, four per thread, all in flight; columns >= n2: copies of the last valid column
        uint4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 512;
            v[i] = *reinterpret_cast<const uint4*>(Bm + (size_t)min(c0 + (e >> 3), n2 - 1) * 64 + (e & 7) * 8);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 512;
            *reinterpret_cast<uint4*>(Dl + (e >> 3) * FT_DS + (e & 7) * 8) = v[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) (&colx[0][0])[tid + i * 512] = -INFINITY;
    }
    f16x8 a[4], an[4];
    int blk = blk_lo + wave, nxt;
    if (blk < nblock) {
        const int row = min(blk * 32 + l31, n1 - 1);       // rows >= n1: copies of the last valid row (never reported)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) an[kk] = *reinterpret_cast<const f16x8*>(A + (size_t)row * 64 + kk * 16 + half * 8);
    }
    __syncthreads();
    float* mycolx = &colx[wave][l31];
    const _Float16* bp0 = Dl + l31 * FT_DS + half * 8;
    while (blk < nblock) {
        {   // the next block of this wave, and its fragments on their way under this block's MFMAs
            int t = 0;
            if (lane == 0) t = atomicAdd(&next_block, 1);
            nxt = __builtin_amdgcn_readfirstlane(t);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) a[kk] = an[kk];
        if (nxt < nblock) {
            const int row = min(nxt * 32 + l31, n1 - 1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) an[kk] = *reinterpret_cast<const f16x8*>(A + (size_t)row * 64 + kk * 16 + half * 8);
        }
        const int myrow = blk * 32 + l31;
        // branch-free stores (a branch would end the basic block and with it the MFMA / VALU interleave below): buffer stores, lanes that
        // must not store carry an out-of-range offset and the hardware drops them; the C row's range check also drops columns >= n2
        const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(C + ((size_t)p * nrb32 + blk) * N2 + c0, 0, (min(n2, c0 + FT_COLS) - c0) * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(R + ((size_t)p * ncb32 + (c0 >> 5)) * N1, 0, FT_TILES * N1 * 2, 0x00020000);
        const int offC = half == 0 ? l31 * 2 : (int)0x80000000;
        const int offR = (half == 1 && myrow < n1) ? myrow * 2 : (int)0x80000000;
        float rowrun = -INFINITY;
        f16x8 bfrag[2][4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) bfrag[0][kk] = *reinterpret_cast<const f16x8*>(bp0 + kk * 16);
        f32x16 acc, accT;
        // Software pipeline over the tiles (uniform control flow only: every tile is one basic block, so the scheduler can interleave):
        //   A(ct): acc  = a . b[ct]   ||  row epilogue of accT(ct - 1)
        //   B(ct): accT = b[ct] . a   ||  column epilogue of acc(ct)
        // with the fragments of tile ct + 1 leaving LDS meanwhile (second register set).
#define XFH_ROW_EPILOGUE(CT)                                                                              \
        {                                                                                                 \
            float rm = max16(accT);                                                                       \
            rm = fmaxf(rm, xhalf(rm));                                                                    \
            rowrun = fmaxf(rowrun, rm);                                                                   \
            __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, f16_up(rm * F16_STORE_SCALE)), rR, offR, (CT) * N1 * 2, 0);       \
        }
#define XFH_INTERLEAVE_4x5()                                                                              \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
#pragma unroll
        for (int ct = 0; ct < FT_TILES; ++ct) {
            if (ct < ntile) {                               // (uniform)
                if (ct + 1 < FT_TILES) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) bfrag[(ct + 1) & 1][kk] = *reinterpret_cast<const f16x8*>(bp0 + (ct + 1) * 32 * FT_DS + kk * 16);
                }
                // ---- A: lane = column c0 + 32 ct + l31, registers = 16 of the block's rows
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kk], bfrag[ct & 1][kk], acc, 0, 0, 0);
                if (ct > 0) XFH_ROW_EPILOGUE(ct - 1)
                XFH_INTERLEAVE_4x5()
                __builtin_amdgcn_sched_barrier(0);
                // ---- B: lane = row myrow, registers = 16 of the tile's columns
#pragma unroll
                for (int r = 0; r < 16; ++r) accT[r] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) accT = __builtin_amdgcn_mfma_f32_32x32x16_f16(bfrag[ct & 1][kk], a[kk], accT, 0, 0, 0);
                {
                    float cm = max16(acc);
                    cm = fmaxf(cm, xhalf(cm));
                    __hip_atomic_fetch_max(mycolx + ct * 32, cm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (both half-waves hold cm)
                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, f16_up(cm * F16_STORE_SCALE)), rC, offC + ct * 64, 0, 0);
                }
                XFH_INTERLEAVE_4x5()
                __builtin_amdgcn_sched_barrier(0);
                XFH_KEEP_FRAGS(bfrag[ct & 1]);
            }
        }
        XFH_ROW_EPILOGUE(ntile - 1)
#undef XFH_ROW_EPILOGUE
#undef XFH_INTERLEAVE_4x5
        if (half == 0 && myrow < n1) atomicMax(&rowmaxh[(size_t)p * N1 + myrow], float_ord(rowrun));      // (no return value: fire and forget)
        blk = nxt;
    }
    // column maxima: across the 8 waves, then across the row shares by 32-bit atomic max (the thresholds follow in mnn_f16_thr_kernel)
    __syncthreads();
    if (tid < FT_COLS && c0 + tid < n2) {
        float k = colx[0][tid];
#pragma unroll
        for (int w = 1; w < 8; ++w) k = fmaxf(k, colx[w][tid]);
        atomicMax(&colmaxh[(size_t)p * N2 + c0 + tid], float_ord(k));
    }
}

// Sweep, ONE orientation (round 6).  The kernel above pays the matrix pipe twice per tile to get both reductions in-lane, and the counters say the pipe is what it waits for
// (busy 72-77 % of its cycles, at the 1.4-1.5 GHz the chip holds under that load).  Here a tile is multiplied once, as accT = b . a (lane = row of the block, registers = 16 of
// the tile's columns), and the column direction is served by what a C block IS: any partition of a column's rows will do for the filter (the refine evaluates all rows of a flagged
// block exactly), so a block is made of the rows that meet in one lane --
//     C block (g, l) of column j = the rows {1024 g + 32 t + l : t = 0..31}         (row group g = 32 consecutive row blocks, residue l = the lane)
// and its maximum is an ELEMENTWISE running maximum over the row blocks a wave walks: 16 v_max_f32 per tile, no reduction at all.  A wave owns one row group (32 row blocks x the
// workgroup's 256 columns = 256 tiles, its 8 x 16 running maxima in registers), the R blocks stay what they were (32 consecutive columns: in-lane max3 tree), and when the group is
// done the running maxima pass through a 32 x 32 LDS tile per column tile: read down the columns they give the column maxima (one atomic per column and wave), read along the rows
// the C rows (fp16, a quarter, rounded up -- as before).  4 MFMAs (128 pipe cycles) and ~ 32 VALU ops per tile.
// Every one of the 8 column tiles is always multiplied (columns >= n2 are copies of the last valid column in LDS; their stores are dropped by range checks): a row block is one basic block.
//   C (P, 32 ceil(N1 / 1024), N2): row g * 32 + l = block (g, l)           R, rowmaxh, colmaxh: as above
constexpr int S2_T = 32;                  // row blocks per row group
constexpr int S2_GROUP = 32 * S2_T;       // rows per row group
constexpr int S2_WAVES = 4;               // row groups (waves) per workgroup
constexpr int S2_TILES = 4;              // column tiles per workgroup (128 columns in LDS): 64 registers of running maxima per wave
constexpr int S2_COLS = 32 * S2_TILES;
constexpr int S2_XP = 33;                 // pitch of a wave's 32 x 32 transposition tile, in floats
__host__ __device__ inline int s2_c_blocks(int N1) { return ceil_div(N1, S2_GROUP) * 32; }
__global__ __launch_bounds__(64 * S2_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2)))
void mnn_f16_sweep2_kernel(const _Float16* __restrict__ a16, size_t sa16, const _Float16* __restrict__ b16, size_t sb16,
                           const int32_t* __restrict__ n1p, const int32_t* __restrict__ n2p, int n_stride, int n_off2,
                           int N1, int N2, int ncc, int ngq, int P,
                           unsigned* __restrict__ colmaxh, unsigned* __restrict__ rowmaxh, _Float16* __restrict__ R, _Float16* __restrict__ C) {
    __shared__ __attribute__((aligned(16))) _Float16 Dl2[S2_COLS * FT_DS];      // the columns
    __shared__ __attribute__((aligned(16))) _Float16 af[S2_WAVES][64 * FT_DS];  // per wave: the 64 rows of a block pair on their way into MFMA operand order (and, after the loop, the transposition tile of the flush)
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int p, item;
    if (!xcd_group_map(blockIdx.x, ncc * ngq, P, p, item)) return;
    const int cc = item / ngq, gq = item - cc * ngq;           // column chunk, and which four row groups
    const int n1 = fpair_count(n1p, p * n_stride, N1);
    const int n2 = fpair_count(n2p, p * n_stride + n_off2, N2);
    const int c0 = cc * S2_COLS;
    if (n1 <= 0 || n2 <= 0 || c0 >= n2 || gq * S2_WAVES * S2_GROUP >= n1) return;
    const _Float16* A = a16 + (size_t)p * sa16;
    const _Float16* Bm = b16 + (size_t)p * sb16;
    const int ncb32 = ceil_div(N2, 32);
    {   // 128 columns x 64 fp16 = 1024 16-byte pieces, four per thread, all in flight; columns >= n2: copies of the last valid column
        constexpr int NP = S2_COLS * 8 / (64 * S2_WAVES);
        uint4 v[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int e = tid + i * (64 * S2_WAVES);
            v[i] = *reinterpret_cast<const uint4*>(Bm + (size_t)min(c0 + (e >> 3), n2 - 1) * 64 + (e & 7) * 8);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int e = tid + i * (64 * S2_WAVES);
            *reinterpret_cast<uint4*>(Dl2 + (e >> 3) * FT_DS + (e & 7) * 8) = v[i];
        }
    }
    const int g = gq * S2_WAVES + wave;
    const int blk_lo = g * S2_T, nblock = min(ceil_div(n1, 32), blk_lo + S2_T);
    // TWO row blocks per step (64 consecutive rows: lanes 0-31 and 32-63 of an A fragment hold the same 32 rows, so block X and block Y each have their own fragments):
    // every B fragment that leaves LDS feeds two MFMAs, and the two accumulators of a tile meet in ONE v_max3 per running maximum
    f16x8 ax[4], ay[4], anx[4], any_[4];
    // The 64 rows of a block pair are 8 KB of consecutive memory.  A lane reading ITS row's 16-byte pieces straight into operand order is a 32-line gather per instruction
    // (the CU's address unit takes one line per cycle: eight such loads per pair and wave were 22 us of this kernel); here a wave requests the 8 KB in eight fully coalesced
    // instructions (lane i of instruction q: piece 64 q + i), passes them through its own LDS rows (the columns' padded layout: conflict-free both ways) and reads its fragments back.
#define XFH_S2_REQUEST(BLK)                                                                                \
    {                                                                                                     \
        /* rows >= n1: copies of the last valid row (never reported); a pair without a second block multiplies the first one twice (no row of another residue gets into a C block) */ \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                   \
            const int e = q * 64 + lane, rx = min((BLK) * 32 + (e >> 3), n1 - 1), ry = (BLK) + 1 < nblock ? min((BLK) * 32 + 32 + (e >> 3), n1 - 1) : rx;       \
            anx[q] = *reinterpret_cast<const f16x8*>(A + (size_t)rx * 64 + (e & 7) * 8);                  \
            any_[q] = *reinterpret_cast<const f16x8*>(A + (size_t)ry * 64 + (e & 7) * 8);                 \
        }                                                                                                 \
    }
    XFH_S2_REQUEST(min(blk_lo, max(nblock - 1, 0)))
    __syncthreads();
    if (blk_lo >= nblock) return;                              // (whole waves leave: nothing below synchronises across waves)
    const _Float16* bp0 = Dl2 + l31 * FT_DS + half * 8;
    _Float16* afw = af[wave];
    f32x16 M[S2_TILES];                                        // running maxima: M[ct][r] = max over this wave's row blocks of S^(32 blk + l31, column (r & 3) + 8 (r >> 2) + 4 half of tile ct)
#pragma unroll
    for (int ct = 0; ct < S2_TILES; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) M[ct][r] = -INFINITY;
    // R rows of this chunk: tiles beyond the last column block of R are outside the resource's range (their stores are dropped)
    const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(R + ((size_t)p * ncb32 + (c0 >> 5)) * N1, 0, min(S2_TILES, ncb32 - (c0 >> 5)) * N1 * 2, 0x00020000);
    // B fragments: a rolling set of four (16 steps per block pair: the ring closes; step = tile ct, K chunk kk; the fragment of step s + 2 leaves LDS while step s is multiplied -- the steps of the next block pair read
    // the same addresses, so the ring never drains)
    constexpr int NSTEP = 4 * S2_TILES;
    f16x8 bq[4];
    bq[0] = *reinterpret_cast<const f16x8*>(bp0);
    bq[1] = *reinterpret_cast<const f16x8*>(bp0 + 16);
    for (int blk = blk_lo; blk < nblock; blk += 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = q * 64 + lane;
            *reinterpret_cast<f16x8*>(afw + (e >> 3) * FT_DS + (e & 7) * 8) = anx[q];
            *reinterpret_cast<f16x8*>(afw + (32 + (e >> 3)) * FT_DS + (e & 7) * 8) = any_[q];
        }
        XFH_WAVE_SYNC();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            ax[kk] = *reinterpret_cast<const f16x8*>(afw + l31 * FT_DS + kk * 16 + half * 8);
            ay[kk] = *reinterpret_cast<const f16x8*>(afw + (32 + l31) * FT_DS + kk * 16 + half * 8);
        }
        XFH_WAVE_SYNC();
        // NO BRANCH in this loop: behind a conditional memory operation hipcc waits for EVERY outstanding one (vmcnt(0)) -- the R stores and the row-maximum atomic of the pair
        // before, a full trip to memory per iteration (the kernel ran at half its speed).  The last pair requests its own fragments again; the atomic goes to the row whose
        // maximum the lane really holds.
        XFH_S2_REQUEST(min(blk + 2, nblock - 1))               // the next pair's fragments on their way under this pair's MFMAs
        const int myrow = blk * 32 + lane;                     // after the half-wave exchange below lane l holds row 32 blk + l of the 64
        const int offR = myrow < n1 ? myrow * 2 : (int)0x80000000;      // (branch-free stores: lanes that must not store carry an out-of-range offset)
        const int arow = half && blk + 1 >= nblock ? min(blk * 32 + l31, n1 - 1) : min(myrow, n1 - 1);      // (a pair without a second block multiplied the first one twice)
        float rowrun = -INFINITY;
        f32x16 accx[2], accy[2];
        // Software pipeline over the tiles, one basic block, PROGRAM ORDER = ISSUE ORDER (scheduling fences between the groups, the values pinned by empty asm statements;
        // left alone, hipcc sank all the elementwise maxima of a row block behind its last MFMA): a tile is 4 K chunks x 2 MFMAs (block X, block Y), and behind each chunk's
        // pair comes a quarter of the reduction of tile ct - 1 (~ 10 vector instructions): the max3 tree of X, the tree of Y, the half-wave exchange (v_permlane32_swap
        // of the two row maxima: lanes 0-31 end up with block X's, lanes 32-63 with block Y's -- one fp16 conversion and ONE 128-byte store for 64 rows) and the running
        // maxima M = max3(M, X, Y).
#define XFH_S2_FENCE() __builtin_amdgcn_sched_barrier(0)
#define XFH_S2_M(CT, R0, R1) { _Pragma("unroll") for (int r = (R0); r < (R1); ++r) M[CT][r] = fmaxf(fmaxf(M[CT][r], accx[(CT) & 1][r]), accy[(CT) & 1][r]); }
#define XFH_S2_PIN2(CT, R0) asm volatile("" : "+v"(M[CT][R0]), "+v"(M[CT][(R0) + 1]))
#define XFH_S2_PIN10(CT, R0) asm volatile("" : "+v"(M[CT][R0]), "+v"(M[CT][(R0) + 1]), "+v"(M[CT][(R0) + 2]), "+v"(M[CT][(R0) + 3]), "+v"(M[CT][(R0) + 4]), "+v"(M[CT][(R0) + 5]), \
                                                "+v"(M[CT][(R0) + 6]), "+v"(M[CT][(R0) + 7]), "+v"(M[CT][(R0) + 8]), "+v"(M[CT][(R0) + 9]))
#define XFH_S2_EPI0(CT) rx_ = max16(accx[(CT) & 1]); XFH_S2_M(CT, 0, 2) XFH_S2_PIN2(CT, 0); asm volatile("" : "+v"(rx_));
#define XFH_S2_EPI1(CT) ry_ = max16(accy[(CT) & 1]); XFH_S2_M(CT, 2, 4) XFH_S2_PIN2(CT, 2); asm volatile("" : "+v"(ry_));
#define XFH_S2_EPI2(CT)                                                                                   \
            {                                                                                             \
                const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(rx_), __float_as_uint(ry_), false, false);      /* {x.lo, y.lo}, {x.hi, y.hi} */ \
                const float rm_ = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));                \
                rowrun = fmaxf(rowrun, rm_);                                                              \
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, f16_up(rm_ * F16_STORE_SCALE)), rR, offR + (CT) * N1 * 2, 0, 0);   /* (in the lane offset: the range check does not see a scalar offset) */ \
            }                                                                                             \
            XFH_S2_M(CT, 4, 6) XFH_S2_PIN2(CT, 4); asm volatile("" : "+v"(rowrun));
#define XFH_S2_EPI3(CT) XFH_S2_M(CT, 6, 16) XFH_S2_PIN10(CT, 6);
        float rx_ = 0.f, ry_ = 0.f;
#pragma unroll
        for (int ct = 0; ct < S2_TILES; ++ct)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int s_ = 4 * ct + kk;
            {   // the fragment two steps ahead
                const int sn = (s_ + 2) % NSTEP;
                bq[(s_ + 2) & 3] = *reinterpret_cast<const f16x8*>(bp0 + (sn >> 2) * 32 * FT_DS + (sn & 3) * 16);
            }
            f32x16& cx = accx[ct & 1];
            f32x16& cy = accy[ct & 1];
            if (kk == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { cx[r] = 0.f; cy[r] = 0.f; }
            }
            cx = __builtin_amdgcn_mfma_f32_32x32x16_f16(bq[s_ & 3], ax[kk], cx, 0, 0, 0);
            cy = __builtin_amdgcn_mfma_f32_32x32x16_f16(bq[s_ & 3], ay[kk], cy, 0, 0, 0);
            asm volatile("" : "+v"(cx), "+v"(cy));
            XFH_S2_FENCE();
            if (ct > 0) {
                if (kk == 0) { XFH_S2_EPI0(ct - 1) }
                if (kk == 1) { XFH_S2_EPI1(ct - 1) }
                if (kk == 2) { XFH_S2_EPI2(ct - 1) }
                if (kk == 3) { XFH_S2_EPI3(ct - 1) }
                XFH_S2_FENCE();
            }
            asm volatile("" :: "v"(bq[s_ & 3]));          // (a fragment stays alive until the group behind its MFMAs is over: tools/check_mfma_war.py)
        }
        {
            XFH_S2_EPI0(S2_TILES - 1)
            XFH_S2_EPI1(S2_TILES - 1)
            XFH_S2_EPI2(S2_TILES - 1)
            XFH_S2_EPI3(S2_TILES - 1)
        }
        XFH_S2_FENCE();
        XFH_KEEP_FRAGS(ax);          // (the next pair's fragments must not be copied over this pair's right behind its last MFMA)
        XFH_KEEP_FRAGS(ay);
#undef XFH_S2_FENCE
#undef XFH_S2_PIN2
#undef XFH_S2_PIN10
#undef XFH_S2_M
#undef XFH_S2_EPI0
#undef XFH_S2_EPI1
#undef XFH_S2_EPI2
#undef XFH_S2_EPI3
        atomicMax(&rowmaxh[(size_t)p * N1 + arow], float_ord(rowrun));      // (no return value: fire and forget; a row >= n1 is a copy of row n1 - 1 and carries that row's maximum)
    }
    // ---- flush: the running maxima of this row group, tile by tile through the wave's 32 x 32 LDS tile (row = residue l, column = column of the tile)
    float* xw = reinterpret_cast<float*>(af[wave]);          // (the fragment rows are done with)
    static_assert(32 * S2_XP * 4 <= 64 * FT_DS * 2, "the flush tile lives in the wave's fragment rows");
    const bool vec_ok = (N2 & 7) == 0;                         // 16-byte stores (eight fp16 values) need 16-byte aligned rows of C
    _Float16* crow = C + ((size_t)p * s2_c_blocks(N1) + g * 32 + l31) * N2 + c0;
#pragma unroll
    for (int ct = 0; ct < S2_TILES; ++ct) {
#pragma unroll
        for (int r = 0; r < 16; ++r) xw[l31 * S2_XP + (r & 3) + 8 * (r >> 2) + 4 * half] = M[ct][r];
        XFH_WAVE_SYNC();
        // down the columns: lane (c, half) takes residues 16 half .. 16 half + 15 of column c
        float cm = xw[(16 * half) * S2_XP + l31];
#pragma unroll
        for (int t = 1; t < 16; ++t) cm = fmaxf(cm, xw[(16 * half + t) * S2_XP + l31]);
        cm = fmaxf(cm, xhalf(cm));
        const int col = c0 + ct * 32 + l31;
        if (half == 0 && col < n2) atomicMax(&colmaxh[(size_t)p * N2 + col], float_ord(cm));
        // along the rows: lane (l, half) takes columns 16 half .. 16 half + 15 of residue l
        const int cs = ct * 32 + 16 * half;
        _Float16 o[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) o[t] = f16_up(xw[l31 * S2_XP + 16 * half + t] * F16_STORE_SCALE);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            f16x8 v;
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = o[8 * q + t];
            if (vec_ok && c0 + cs + 8 * q + 8 <= n2) *reinterpret_cast<f16x8*>(crow + cs + 8 * q) = v;
            else {
#pragma unroll
                for (int t = 0; t < 8; ++t)
                    if (c0 + cs + 8 * q + t < n2) crow[cs + 8 * q + t] = v[t];
            }
        }
        XFH_WAVE_SYNC();
    }
}

// thresholds, once the maxima are complete: thr = max^ - 2 E (scaled units), stored like the block maxima (a quarter, fp16) but rounded DOWN.  grid (ceil(max(N1,N2)/256), P, 2 sides)
__global__ __launch_bounds__(256) void mnn_f16_thr_kernel(float unit_bound, const int32_t* __restrict__ n1p, const int32_t* __restrict__ n2p, int n_stride, int n_off2,
                                                          int N1, int N2, int P, const float* __restrict__ na, const float* __restrict__ nb,
                                                          const unsigned* __restrict__ nmax, const unsigned* __restrict__ rowmaxh, const unsigned* __restrict__ colmaxh,
                                                          _Float16* __restrict__ thr_row, _Float16* __restrict__ thr_col) {
    const int p = blockIdx.y, side = blockIdx.z, i = blockIdx.x * 256 + threadIdx.x;
    const int N = side ? N2 : N1;
    if (i >= (side ? fpair_count(n2p, p * n_stride + n_off2, N2) : fpair_count(n1p, p * n_stride, N1))) return;
    const PairWindow w = pair_window(unit_bound, nmax, P, p, side == 0);
    const float nx = unit_bound > 0.f ? unit_bound : (side ? nb : na)[(size_t)p * N + i];
    (side ? thr_col : thr_row)[(size_t)p * N + i] = f16_down((ord_float((side ? colmaxh : rowmaxh)[(size_t)p * N + i]) - fmaf(w.two_c, nx, w.two_k)) * F16_STORE_SCALE);
}

constexpr int RF_ROUND = 512;      // x per scan round
constexpr int RF_CHUNK = 4096;     // x per pass over the block maxima
constexpr int RF_QUEUE = 1024;     // queue capacity per wave; it is drained before a scan round (<= 512 new entries) could overflow it
constexpr int RF_XS = 68;          // LDS row stride of the 32 x 64 tile in floats (272 bytes = 17 x 16: conflict-free 16-byte reads down a column of rows)
constexpr int RF_WAVES = 2;        // waves (= Y blocks) per workgroup
__global__ __launch_bounds__(64 * RF_WAVES) __attribute__((amdgpu_waves_per_eu(4, 8))) void mnn_f16_refine_kernel(
    const float* __restrict__ d1, size_t ps1, const float* __restrict__ d2, size_t ps2, const int32_t* __restrict__ n1p,
    const int32_t* __restrict__ n2p, int n_stride, int n_off2, int N1, int N2, int nyb_max, int c_strided, int P,
    const _Float16* __restrict__ thr_row, const _Float16* __restrict__ thr_col, const _Float16* __restrict__ R,
    const _Float16* __restrict__ C, unsigned long long* __restrict__ rowkey, unsigned long long* __restrict__ colkey) {
    __shared__ unsigned short queue[RF_WAVES][RF_QUEUE];                       // per wave: flagged x (offsets into the current chunk)
    __shared__ __attribute__((aligned(16))) float tile[RF_WAVES][32 * RF_XS];   // per wave: 32 rows x 64 channels on their way into MFMA operand order
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int p, item;
    if (!xcd_group_map(blockIdx.x, 2 * (nyb_max / RF_WAVES), P, p, item)) return;      // nyb_max is a multiple of RF_WAVES
    const int side = item / (nyb_max / RF_WAVES), yb = (item - side * (nyb_max / RF_WAVES)) * RF_WAVES + wv;
    const int n1 = fpair_count(n1p, p * n_stride, N1);
    const int n2 = fpair_count(n2p, p * n_stride + n_off2, N2);
    if (n1 <= 0 || n2 <= 0) return;
    const int nX = side ? n2 : n1, nY = side ? n1 : n2, NX = side ? N2 : N1;
    // the 32 Y rows of block yb: consecutive (R blocks; C blocks of the two-orientation sweep), or -- C blocks of the one-orientation sweep -- the rows of residue yb & 31 in row group yb >> 5
    const bool strided = side && c_strided;                 // (uniform)
    const int ybase = strided ? (yb >> 5) * S2_GROUP + (yb & 31) : yb * 32, ystep = strided ? 32 : 1;
    if (ybase >= nY) return;                                // (whole waves leave: nothing below synchronises across waves)
    const float* X = side ? d2 + (size_t)p * ps2 : d1 + (size_t)p * ps1;
    const float* Y = side ? d1 + (size_t)p * ps1 : d2 + (size_t)p * ps2;
    const _Float16* M = (side ? C + (size_t)p * (c_strided ? s2_c_blocks(N1) : ceil_div(N1, 32)) * N2 : R + (size_t)p * ceil_div(N2, 32) * N1) + (size_t)yb * NX;
    const _Float16* T = side ? thr_col + (size_t)p * N2 : thr_row + (size_t)p * N1;
    unsigned long long* key = side ? colkey + (size_t)p * N2 : rowkey + (size_t)p * N1;
    const bool vec_ok = (NX & 7) == 0;                      // 16-byte loads (eight fp16 values) need 16-byte aligned rows of M / thresholds
    unsigned short* q = queue[wv];
    float* xt = tile[wv];
    // buffer resources (uniform bases in SGPRs, 32-bit lane offsets; reads past the end return 0): one address register per lane instead of
    // a 64-bit pointer per array
    const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(M), 0, nX * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rT = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(T), 0, nX * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, nX * 256, 0x00020000);
    const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Y), 0, nY * 256, 0x00020000);

    // 32 rows of 256 bytes -> LDS -> this lane's 32 channels (32 half ..) of row l31: the global reads are coalesced (16 lanes per row, four
    // rows per instruction); a lane reading its own row straight from memory is a 64-line gather per instruction, and those gathers were
    // what the matrix-core version of this kernel waited for
    const int trow = lane >> 4, tq = lane & 15;
    auto rows_to_operand = [&](const __amdgpu_buffer_rsrc_t& rs, auto row_of, float (&reg)[32]) {
        uint4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, row_of(4 * i + trow) * 256 + tq * 16, 0, 0));
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(xt + (4 * i + trow) * RF_XS + tq * 4) = v[i];
        // (wave-synchronous: LDS operations of one wave execute in order)
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float4 f = *reinterpret_cast<const float4*>(xt + l31 * RF_XS + half * 32 + t * 4);
            reg[4 * t] = f.x; reg[4 * t + 1] = f.y; reg[4 * t + 2] = f.z; reg[4 * t + 3] = f.w;
        }
    };
    float yreg[32];                                         // Y row yb*32 + l31, channels 32 half .. ; rows past nY read 0 and are masked below
    rows_to_operand(rY, [&](int r) { return ybase + r * ystep; }, yreg);
    const bool ragged = ybase + 31 * ystep >= nY;           // (uniform) a block with rows past nY: they must not win

    for (int xc = 0; xc < nX; xc += RF_CHUNK) {
        int cnt = 0;
        // ---- phase 2 (called when the queue fills up, and after the scan): up to 32 queued x per pass against the 32 rows of the Y block
        //      on the f32 matrix cores, with the operand roles and the K order of the exact kernel (k_match.hip): step s multiplies channels
        //      s (lanes 0-31) and 32 + s (lanes 32-63) -- the same fma chain, so the refine's similarities are the exact kernel's, bit for
        //      bit.  A = Y rows (i = y), B = x rows (j = queue entry): a lane ends up with 16 of the 32 similarities of ITS entry; maximum
        //      and first index in-lane, the other half by permlane32.
        auto drain = [&]() {
            for (int e0 = 0; e0 < cnt; e0 += 32) {
                float xreg[32];
                rows_to_operand(rX, [&](int r) { return xc + (int)q[min(e0 + r, cnt - 1)]; }, xreg);      // (entries past the queue: copies of the last one, not reported)
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int s_ = 0; s_ < 32; ++s_) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(yreg[s_], xreg[s_], acc, 0, 0, 0);
                // acc[r] = S(x of entry l31, y = ybase + ((r&3) + 8*(r>>2) + 4*half) * ystep): y ascends with r
                if (ragged) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (ybase + ((r & 3) + 8 * (r >> 2) + 4 * half) * ystep >= nY) acc[r] = -INFINITY;
                }
                const float m = max16(acc);
                int ry = 0;
#pragma unroll
                for (int r = 15; r >= 0; --r)
                    if (acc[r] == m) ry = (r & 3) + 8 * (r >> 2);                  // the FIRST r attaining the maximum: the lowest y of this half
                unsigned long long k = ((unsigned long long)float_ord(m) << 32) | (0xffffffffu - (unsigned)(ybase + (ry + 4 * half) * ystep));
                k = u64_max(k, xhalf_u64(k));                                        // ties between the halves: the larger ~y = the lower y
                const int e = e0 + l31;
                if (half == 0 && e < cnt) atomicMax(&key[xc + q[e]], k);
            }
            cnt = 0;
        };
        // ---- phase 1: scan.  Eight block maxima per lane and round (ONE 16-byte load of each array), FOUR rounds in flight, every load UNCONDITIONAL (a round past the
        //      end reads zeros through the resource's range check and is not tested): behind a conditional load hipcc waits for every outstanding one (vmcnt(0)), which made
        //      the two-deep prefetch this loop had a chain of eight full memory round trips.  One compare per element, the queue bookkeeping only for the (rare) hits.
        const int nround = ceil_div(min(RF_CHUNK, nX - xc), RF_ROUND);
        auto issue = [&](auto vec, int r, f16x8& m, f16x8& t) {      // vec: std::true_type / false_type -- the choice is made ONCE, outside the rounds (no branch between their loads)
            const int xb = xc + r * RF_ROUND + lane * 8;
            if constexpr (decltype(vec)::value) {
                m = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rM, xb * 2, 0, 0));
                t = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rT, xb * 2, 0, 0));
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    m[j] = __builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(rM, (xb + j) * 2, 0, 0));
                    t[j] = __builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(rT, (xb + j) * 2, 0, 0));
                }
            }
        };
        auto test = [&](int r, const f16x8& m, const f16x8& t) {
            if (cnt > RF_QUEUE - RF_ROUND) drain();          // (uniform; identical descriptor sets get here)
            const int xl = r * RF_ROUND + lane * 8;
            const int left = nX - (xc + xl);                 // elements of this lane that exist (entries past nX read 0 >= 0: mask them)
            // flagged: m >= t, i.e. the SIGN of m - t (four v_pk_add_f16 for the eight values; a difference of fp16 numbers never rounds across zero, and a flushed
            // tiny one keeps its sign: -0 = below the threshold)
            const f16x8 d = m - t;
            const uint4 db = __builtin_bit_cast(uint4, d);
            const unsigned dw[4] = {db.x, db.y, db.z, db.w};
            // ordered compaction into the wave's queue: ballot per j, prefix by popcount
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool neg = (dw[j >> 1] >> (15 + 16 * (j & 1))) & 1u;
                const unsigned long long bal = __ballot(!neg && j < left);
                if (bal) {                                   // (uniform)
                    if ((bal >> lane) & 1ull) q[cnt + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)(xl + j);
                    cnt += __popcll(bal);
                }
            }
        };
        auto scan = [&](auto vec) {
            constexpr int NB = decltype(vec)::value ? 4 : 1;      // (unaligned rows: sixteen 2-byte loads per round, one round at a time)
            for (int r0 = 0; r0 < nround; r0 += NB) {
                f16x8 m4[NB], t4[NB];
#pragma unroll
                for (int k = 0; k < NB; ++k) issue(vec, r0 + k, m4[k], t4[k]);
#pragma unroll
                for (int k = 0; k < NB; ++k)
                    if (r0 + k < nround) test(r0 + k, m4[k], t4[k]);
            }
        };
        if (vec_ok) scan(std::true_type{});
        else scan(std::false_type{});
        drain();
    }
}

// d1_16 / d2_16 (optional, both or neither): fp16 copies the caller already holds (xfh_detect_sparse's desc_f16 = RNE(256 * row)), laid out
// like d1 / d2 (same pair strides in elements), rows L2-normalised: |row| <= 1.00001.  They replace the prep passes.
void launch_match_f16(const MatchWs& ws, const float* d1, size_t ps1, const float* d2, size_t ps2, const uint16_t* d1_16, const uint16_t* d2_16,
                      const int32_t* n1, const int32_t* n2, int n_stride, int n_off2, int P, int N1, int N2, hipStream_t st, Profiler* prof, int sweep_form) {
    const bool prepared = d1_16 && d2_16;
    const _Float16* a16 = prepared ? reinterpret_cast<const _Float16*>(d1_16) : ws.a16;
    const _Float16* b16 = prepared ? reinterpret_cast<const _Float16*>(d2_16) : ws.b16;
    const size_t sa = prepared ? ps1 : (size_t)N1 * 64, sb = prepared ? ps2 : (size_t)N2 * 64;
    const float ub = prepared ? 1.00001f : 0.f;
    if (!prepared) {
        prof_begin(prof, XFH_SPAN_MATCH_PREP, st);
        const dim3 g(ceil_div(N1 > N2 ? N1 : N2, 256), P, 2);
        mnn_prep_kernel<false><<<g, 256, 0, st>>>(d1, ps1, d2, ps2, n1, n2, n_stride, n_off2, N1, N2, ws.a16, ws.b16, ws.na, ws.nb, ws.nmax);
        mnn_prep_kernel<true><<<g, 256, 0, st>>>(d1, ps1, d2, ps2, n1, n2, n_stride, n_off2, N1, N2, ws.a16, ws.b16, ws.na, ws.nb, ws.nmax);
        prof_end(prof, XFH_SPAN_MATCH_PREP, st, 0, 0);
    }
    prof_begin(prof, XFH_SPAN_MATCH_SWEEP, st);
    // the one-orientation sweep (a wave per row group of 1024 rows and column chunk) when its wave tasks fill at least half of the chip's wave slots (two per SIMD);
    // the two-orientation sweep below shares the row blocks out more finely: few pairs, short lists
    const int ngroup = ceil_div(N1, S2_GROUP), ncc2 = ceil_div(N2, S2_COLS);
    const bool one = sweep_form == 2 || ((long)P * ncc2 * ngroup >= 8L * num_cus() && sweep_form != 1);
    if (one) {
        const int ngq = ceil_div(ngroup, S2_WAVES);
        mnn_f16_sweep2_kernel<<<xcd_grid_size(ncc2 * ngq, P), 64 * S2_WAVES, 0, st>>>(a16, sa, b16, sb, n1, n2, n_stride, n_off2, N1, N2, ncc2, ngq, P,
                                                                                    ws.colmaxh, ws.rowmaxh, reinterpret_cast<_Float16*>(ws.R), reinterpret_cast<_Float16*>(ws.C));
    } else {
    const int ncc = ceil_div(N2, FT_COLS);
    // two workgroups per CU fill the chip; with few pairs the row blocks of a column chunk are shared out over more workgroups (>= 8 blocks each)
    const int nsplit = max(1, min(ceil_div(2 * num_cus(), ncc * P), ceil_div(N1, 256)));
    mnn_f16_sweep_kernel<<<xcd_grid_size(ncc * nsplit, P), 512, 0, st>>>(a16, sa, b16, sb, n1, n2, n_stride, n_off2, N1, N2, ncc, nsplit, P,
                                                                        ws.colmaxh, ws.rowmaxh, reinterpret_cast<_Float16*>(ws.R), reinterpret_cast<_Float16*>(ws.C)
#if XFH_CODE_SHIFT > 0
                                                                        , g_debug_cold
#endif
                                                                        );
    }
    prof_end(prof, XFH_SPAN_MATCH_SWEEP, st, 0, 0);
    const int nyb = ceil_div(max(ceil_div(N2, 32), one ? s2_c_blocks(N1) : ceil_div(N1, 32)), RF_WAVES) * RF_WAVES;
    prof_begin(prof, XFH_SPAN_MATCH_REFINE, st);
    mnn_f16_thr_kernel<<<dim3(ceil_div(N1 > N2 ? N1 : N2, 256), P, 2), 256, 0, st>>>(ub, n1, n2, n_stride, n_off2, N1, N2, P, ws.na, ws.nb, ws.nmax, ws.rowmaxh, ws.colmaxh,
                                                                                        reinterpret_cast<_Float16*>(ws.thr_row), reinterpret_cast<_Float16*>(ws.thr_col));
    mnn_f16_refine_kernel<<<xcd_grid_size(2 * (nyb / RF_WAVES), P), 64 * RF_WAVES, 0, st>>>(d1, ps1, d2, ps2, n1, n2, n_stride, n_off2, N1, N2, nyb, one ? 1 : 0, P,
                                                                          reinterpret_cast<const _Float16*>(ws.thr_row), reinterpret_cast<const _Float16*>(ws.thr_col), reinterpret_cast<const _Float16*>(ws.R),
                                                                          reinterpret_cast<const _Float16*>(ws.C), ws.rowkey, ws.colkey);
    prof_end(prof, XFH_SPAN_MATCH_REFINE, st, 0, 0);
}

}  // namespace xfh
