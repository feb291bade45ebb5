// Mutual-nearest-neighbour matching on the f32 matrix cores.
//   XFeat.match        modules/xfeat.py:327-348     XFeat.batch_match  modules/xfeat.py:265-290
//
// The reference materialises S = D1.D2^T AND S^T (two GEMMs, 2 x 67 MB at 4096 points) and
// arg-maxes both.  Here S is produced tile by tile by v_mfma_f32_32x32x2_f32 and never leaves
// registers: every workgroup owns 256 rows of D1 (A fragments stationary in VGPRs, K = 64 is
// only 32 MFMA steps) and sweeps all columns of D2 staged through LDS, keeping a running
// row max/arg-max per lane and emitting per-row-block column maxima as packed 64-bit keys
// (ord(sim) << 32 | ~index), so that ties resolve to the LOWEST index like torch.max.
// A second small kernel reduces the column partials, applies the mutual test (+ optional
// min_cossim) and compacts the surviving pairs in ascending row order.
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
                                                      int N2, int nrb, int P, int* __restrict__ match12,
                                                      float* __restrict__ rowmax, unsigned long long* __restrict__ colpart) {
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
                colpart[((size_t)p * nrb + rb) * N2 + col] = k;
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
        if (l31 == 0 && row < n1) {
            match12[(size_t)p * N1 + row] = (int)(0xffffffffu - (unsigned)(key & 0xffffffffu));
            rowmax[(size_t)p * N1 + row] = ord_float((unsigned)(key >> 32));
        }
    }
}

// grid (P), block 1024, dynamic LDS = N2 ints
__global__ __launch_bounds__(1024) void mnn_finalize_kernel(const int32_t* __restrict__ n1p, const int32_t* __restrict__ n2p,
                                                            int n_stride, int n_off2, int N1, int N2, int nrb,
                                                            const int* __restrict__ match12, const float* __restrict__ rowmax,
                                                            const unsigned long long* __restrict__ colpart, float min_cossim,
                                                            int64_t* __restrict__ idx0, int64_t* __restrict__ idx1,
                                                            int32_t* __restrict__ n_matches) {
    extern __shared__ int m21[];
    __shared__ int wsum[16];
    __shared__ int s_base;
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n1 = pair_count(n1p, p * n_stride, N1);
    const int n2 = pair_count(n2p, p * n_stride + n_off2, N2);
    if (n1 <= 0 || n2 <= 0) {
        if (tid == 0) n_matches[p] = 0;
        return;
    }
    const int nrbp = ceil_div(n1, MT_ROWS);
    for (int col = tid; col < n2; col += 1024) {
        unsigned long long best = 0ull;
        for (int rb = 0; rb < nrbp; ++rb) best = u64_max(best, colpart[((size_t)p * nrb + rb) * N2 + col]);
        m21[col] = (int)(0xffffffffu - (unsigned)(best & 0xffffffffu));
    }
    if (tid == 0) s_base = 0;
    __syncthreads();
    const int* m12 = match12 + (size_t)p * N1;
    const float* rm = rowmax + (size_t)p * N1;
    for (int base = 0; base < n1; base += 1024) {
        const int row = base + tid;
        bool keep = false;
        int m = 0;
        if (row < n1) {
            m = m12[row];
            keep = (m21[m] == row) && (min_cossim <= 0.f || rm[row] > min_cossim);
        }
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int off = s_base, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const int s = wsum[w];
            if (w < wave) off += s;
            tot += s;
        }
        if (keep) {
            idx0[(size_t)p * N1 + off + before] = row;
            idx1[(size_t)p * N1 + off + before] = m;
        }
        __syncthreads();
        if (tid == 0) s_base += tot;
        __syncthreads();
    }
    if (tid == 0) n_matches[p] = s_base;
}

int match_debug_occupancy() {
    int n = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(mnn_sim_kernel), 512, 0);
    return n;
}

void prof_begin(Profiler* p, int which, hipStream_t st);
void prof_end(Profiler* p, int which, hipStream_t st, double flops, double bytes);

void launch_match(const MatchWs& ws, const float* d1, size_t ps1, const float* d2, size_t ps2, const int32_t* n1,
                  const int32_t* n2, int n_stride, int n_off2, int P, int N1, int N2, float min_cossim, int64_t* idx0,
                  int64_t* idx1, int32_t* n_matches, hipStream_t st, Profiler* prof) {
    const int nrb = match_row_blocks(N1);
    prof_begin(prof, 2, st);
    mnn_sim_kernel<<<xcd_grid_size(nrb, P), 512, 0, st>>>(d1, ps1, d2, ps2, n1, n2, n_stride, n_off2, N1, N2, nrb, P, ws.match12,
                                                 ws.rowmax, ws.colpart);
    prof_end(prof, 2, st, 2.0 * P * (double)N1 * N2 * 64, (double)P * (N1 + N2) * 64 * 4);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mnn_finalize_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            16384 * 4);
        attr_set = true;
    }
    mnn_finalize_kernel<<<P, 1024, (size_t)N2 * sizeof(int), st>>>(n1, n2, n_stride, n_off2, N1, N2, nrb, ws.match12,
                                                                  ws.rowmax, ws.colpart, min_cossim, idx0, idx1,
                                                                  n_matches);
}

}  // namespace xfh
