// Mutual-nearest-neighbour matching on the f32 matrix cores: the EXACT kernel.  xfh_match_mnn runs the filter-and-refine path of
// k_match_f16.hip; this kernel is the every-pair reference it is tested against (xfh_set_option(h, "match_exact", 1));
// the finalize kernel at the bottom is shared by both.
//   XFeat.match        modules/xfeat.py:327-348     XFeat.batch_match  modules/xfeat.py:265-290
//
// The reference materialises S = D1.D2^T AND S^T (two GEMMs, 2 x 67 MB at 4096 points) and
// arg-maxes both.  Here S is produced tile by tile by v_mfma_f32_32x32x2_f32 and never leaves
// registers: every workgroup owns 256 rows of D1 (A fragments stationary in VGPRs, K = 64 is
// only 32 MFMA steps) and sweeps all columns of D2 staged through LDS, keeping a running
// row max/arg-max per lane and folding the column maxima of its rows into one packed 64-bit key
// per column (ord(sim) << 32 | ~index, ties resolve to the LOWEST index like torch.max) with a
// fire-and-forget 64-bit atomic max in L2 (the 16 row blocks of a pair meet there; no per-row-block
// partial array, no reduction pass).  A second small kernel applies the mutual test (+ optional
// min_cossim) and compacts the surviving pairs in ascending row order, 1024 rows per workgroup.
#include "../../include/xfeat_hip.h"
#include "kernels.hpp"

namespace xfh {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MT_ROWS = 256;   // rows of D1 per workgroup (8 waves x 32)
constexpr int MT_COLS = 128;   // columns of D2 per LDS fill
constexpr int MT_DS = 68;      // LDS row stride in floats: 16-B aligned rows, conflict-free ds_read_b128

int match_row_blocks(int N1) { return ceil_div(N1, MT_ROWS); }

__device__ inline int pair_count(const int32_t* n, int idx, int cap) {
    if (!n) return cap;
    const int v = n[idx];
    return v < 0 ? 0 : (v > cap ? cap : v);
}

// 512 threads = 8 waves, 32 rows of D1 each (two waves per SIMD: while one wave runs its VALU
// arg-max epilogue the other keeps the matrix pipe busy).
__global__ __launch_bounds__(512, 4) void mnn_sim_kernel(const float* __restrict__ d1, size_t ps1, const float* __restrict__ d2,
                                                      size_t ps2, const int32_t* __restrict__ n1p,
                                                      const int32_t* __restrict__ n2p, int n_stride, int n_off2, int N1,
                                                      int N2, int nrb, int P, unsigned long long* __restrict__ rowkey,
                                                      unsigned long long* __restrict__ colbest_g) {
    __shared__ __attribute__((aligned(16))) float Dl[MT_COLS * MT_DS];
    __shared__ unsigned long long colbest[8][MT_COLS];

    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // all row blocks of a pair on one XCD: its D2 (1 MB at 4096 points) stays in that XCD's L2
    int p, rb;
    if (!xcd_group_map(blockIdx.x, nrb, P, p, rb)) return;
    const int n1 = pair_count(n1p, p * n_stride, N1);
    const int n2 = pair_count(n2p, p * n_stride + n_off2, N2);
    const int row0 = rb * MT_ROWS;
    if (n1 <= 0 || n2 <= 0 || row0 >= n1) return;
    const float* A = d1 + (size_t)p * ps1;
    const float* Bm = d2 + (size_t)p * ps2;
    const int wrow0 = row0 + wave * 32;

    // stationary A fragment (32 rows x K=64): step s uses k = s (lanes 0-31) / k = 32+s (lanes 32-63)
    float a[32];
    {
        const int row = min(wrow0 + l31, n1 - 1);
        const float4* src = reinterpret_cast<const float4*>(A + (size_t)row * 64 + 32 * half);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 v = src[q];
            a[4 * q + 0] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
        }
    }
    float bv[16];
    int bc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { bv[r] = -INFINITY; bc[r] = 0; }

    for (int c0 = 0; c0 < n2; c0 += MT_COLS) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 512;
            const int col = e >> 4, q = e & 15;
            const int gc = min(c0 + col, n2 - 1);
            const float4 v = *reinterpret_cast<const float4*>(Bm + (size_t)gc * 64 + 4 * q);
            *reinterpret_cast<float4*>(Dl + col * MT_DS + 4 * q) = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int ct = 0; ct < MT_COLS / 32; ++ct) {
            const int cbase = c0 + ct * 32;
            if (cbase >= n2) break;
            // B fragment in two halves of 16 k-steps: 16 VGPRs live instead of 32 (the other three
            // waves of the SIMD cover the second half's LDS latency)
            const float4* bp = reinterpret_cast<const float4*>(Dl + (ct * 32 + l31) * MT_DS + 32 * half);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
                float bf[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = bp[hq * 4 + q];
                    bf[4 * q + 0] = v.x; bf[4 * q + 1] = v.y; bf[4 * q + 2] = v.z; bf[4 * q + 3] = v.w;
                }
#pragma unroll
                for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[hq * 16 + s], bf[s], acc, 0, 0, 0);
            }
            // D[i=row][j=col]: this lane holds column cbase+l31, rows (r&3)+8*(r>>2)+4*half.
            // No validity masks in this loop: rows >= n1 and columns >= n2 were loaded as copies of the last valid row /
            // column, so they tie with it and lose every first-index tie-break (their indices are larger); the writes
            // of match12 / colpart skip them.
            const int col = cbase + l31;
            // row direction: running max per (lane,row); strict > keeps the earliest column
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[r];
                if (v > bv[r]) { bv[r] = v; bc[r] = col; }
            }
            // column direction: float max over this lane's 16 rows, then the FIRST row attaining it
            // (rows ascend with r); one packed key per lane per tile
            float cm = acc[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) cm = fmaxf(cm, acc[r]);
            int crow = 0x7fffffff;
#pragma unroll
            for (int r = 15; r >= 0; --r)
                if (acc[r] == cm) crow = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            unsigned long long best = ((unsigned long long)float_ord(cm) << 32) | (0xffffffffu - (unsigned)crow);
            best = u64_max(best, xhalf_u64(best));
            if (half == 0) colbest[wave][ct * 32 + l31] = best;
        }
        __syncthreads();
        if (tid < MT_COLS) {
            const int col = c0 + tid;
            if (col < n2) {
                unsigned long long k = colbest[0][tid];
#pragma unroll
                for (int w = 1; w < 8; ++w) k = u64_max(k, colbest[w][tid]);
                atomicMax(&colbest_g[(size_t)p * N2 + col], k);            // no return value: global_atomic_umax_x2, asynchronous
            }
        }
    }

    // row arg-max: reduce the per-lane running maxima over the 32 lanes that share the rows
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        unsigned long long key = ((unsigned long long)float_ord(bv[r]) << 32) | (0xffffffffu - (unsigned)bc[r]);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) key = u64_max(key, shfl_xor_u64(key, o));
        const int row = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (l31 == 0 && row < n1) rowkey[(size_t)p * N1 + row] = key;
    }
}

// grid (P * chunks), block 1024: workgroup (p, q) owns rows [1024 q, 1024 q + 1024) of pair p.  The output position of
// a kept row is the number of kept rows before it: the workgroup recounts the rows of the chunks in front of its own
// (<= 3 cheap passes at N1 = 4096: two 4-byte loads and one 8-byte gather per row) instead of waiting for them.
__device__ inline bool mutual_keep(const unsigned long long* __restrict__ rk, const unsigned long long* __restrict__ cb, int row, int n2,
                                   float min_cossim, int& m) {
    const unsigned long long key = rk[row];                                        // (ord(row max) << 32) | ~arg-max column
    m = (int)(0xffffffffu - (unsigned)(key & 0xffffffffu));
    if (key == 0ull || (unsigned)m >= (unsigned)n2) { m = 0; return false; }       // no key was ever folded in (non-finite descriptors): no match, no out-of-range read
    const int back = (int)(0xffffffffu - (unsigned)(cb[m] & 0xffffffffu));        // arg-max row of column m
    return (back == row) && (min_cossim <= 0.f || ord_float((unsigned)(key >> 32)) > min_cossim);
}

__global__ __launch_bounds__(1024) void mnn_finalize_kernel(const int32_t* __restrict__ n1p, const int32_t* __restrict__ n2p,
                                                            int n_stride, int n_off2, int N1, int N2, int chunks,
                                                            const unsigned long long* __restrict__ rowkey,
                                                            const unsigned long long* __restrict__ colbest_g, float min_cossim,
                                                            int64_t* __restrict__ idx0, int64_t* __restrict__ idx1,
                                                            int32_t* __restrict__ n_matches) {
    __shared__ int wsum[16];
    __shared__ int s_before;
    const int p = blockIdx.x / chunks, q = blockIdx.x - p * chunks;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n1 = pair_count(n1p, p * n_stride, N1);
    const int n2 = pair_count(n2p, p * n_stride + n_off2, N2);
    if (n1 <= 0 || n2 <= 0) {
        if (q == 0 && tid == 0) n_matches[p] = 0;
        return;
    }
    if (q * 1024 >= n1) return;
    const unsigned long long* rk = rowkey + (size_t)p * N1;
    const unsigned long long* cb = colbest_g + (size_t)p * N2;
    // kept rows in the chunks before this one
    int cnt = 0;
    for (int row = tid; row < q * 1024; row += 1024) {
        int m;
        cnt += mutual_keep(rk, cb, row, n2, min_cossim, m) ? 1 : 0;
    }
    cnt = wave_sum_i(cnt);
    if (tid == 0) s_before = 0;
    __syncthreads();
    if (lane == 0 && cnt) atomicAdd(&s_before, cnt);
    __syncthreads();
    // own chunk: ordered compaction
    const int row = q * 1024 + tid;
    int m = 0;
    const bool keep = row < n1 && mutual_keep(rk, cb, row, n2, min_cossim, m);
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int off = s_before, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int sv = wsum[w];
        if (w < wave) off += sv;
        tot += sv;
    }
    if (keep) {
        const int o = off + __popcll(bal & ((1ull << lane) - 1ull));
        idx0[(size_t)p * N1 + o] = row;
        idx1[(size_t)p * N1 + o] = m;
    }
    if (tid == 0 && (q + 1) * 1024 >= n1) n_matches[p] = s_before + tot;       // the pair's last chunk knows the total
}

int match_debug_occupancy() {
    int n = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(mnn_sim_kernel), 512, 0);
    return n;
}

void prof_begin(Profiler* p, int which, hipStream_t st);
void prof_end(Profiler* p, int which, hipStream_t st, double flops, double bytes);

void launch_match_f16(const MatchWs& ws, const float* d1, size_t ps1, const float* d2, size_t ps2, const uint16_t* d1_16, const uint16_t* d2_16,
                      const int32_t* n1, const int32_t* n2, int n_stride, int n_off2, int P, int N1, int N2, hipStream_t st, Profiler* prof, int sweep_form);

void launch_match(const MatchWs& ws, const float* d1, size_t ps1, const float* d2, size_t ps2, const int32_t* n1,
                  const int32_t* n2, int n_stride, int n_off2, int P, int N1, int N2, float min_cossim, int64_t* idx0,
                  int64_t* idx1, int32_t* n_matches, hipStream_t st, Profiler* prof, const uint16_t* d1_16, const uint16_t* d2_16, bool exact_only, int sweep_form) {
    const int nrb = match_row_blocks(N1);
    prof_begin(prof, XFH_SPAN_MATCH_ZERO, st);
    (void)hipMemsetAsync(ws.zeroed, 0, ws.zeroed_bytes, st);   // keys / maxima: 0 = below everything
    prof_end(prof, XFH_SPAN_MATCH_ZERO, st, 0, 0);
    prof_begin(prof, XFH_PROF_MATCH, st);
    if (!exact_only) launch_match_f16(ws, d1, ps1, d2, ps2, d1_16, d2_16, n1, n2, n_stride, n_off2, P, N1, N2, st, prof, sweep_form);
    else {
        prof_begin(prof, XFH_SPAN_MATCH_EXACT, st);
        mnn_sim_kernel<<<xcd_grid_size(nrb, P), 512, 0, st>>>(d1, ps1, d2, ps2, n1, n2, n_stride, n_off2, N1, N2, nrb, P, ws.rowkey, ws.colkey);
        prof_end(prof, XFH_SPAN_MATCH_EXACT, st, 0, 0);
    }
    prof_end(prof, XFH_PROF_MATCH, st, 2.0 * P * (double)N1 * N2 * 64, (double)P * (N1 + N2) * 64 * 4);
    const int chunks = ceil_div(N1, 1024);
    prof_begin(prof, XFH_SPAN_MATCH_FINALIZE, st);
    mnn_finalize_kernel<<<P * chunks, 1024, 0, st>>>(n1, n2, n_stride, n_off2, N1, N2, chunks, ws.rowkey, ws.colkey,
                                                     min_cossim, idx0, idx1, n_matches);
    prof_end(prof, XFH_SPAN_MATCH_FINALIZE, st, 0, 0);
}

}  // namespace xfh
