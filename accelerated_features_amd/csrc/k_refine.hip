// Match refinement epilogue (the MLP itself runs on linear_mfma_kernel):
//   XFeat.refine_matches     modules/xfeat.py:306-325
//   XFeat.subpix_softmax2d   modules/xfeat.py:292-304
// Matches of all P pairs are packed into one compact row list (row -> p*N + r) so the
// fine_matcher GEMMs only touch live rows; the confidence filter keeps match order per pair.
#include "kernels.hpp"

namespace xfh {

__device__ inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// one block: exclusive scan of clamp(n_matches[p], 0, N) -> offs[0..P], total = offs[P]
__global__ __launch_bounds__(1024) void refine_offsets_kernel(const int32_t* __restrict__ n_matches, int P, int N,
                                                              int32_t* __restrict__ offs, int32_t* __restrict__ total) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < P; base += 1024) {
        const int p = base + tid;
        const int v = p < P ? clampi(n_matches[p], 0, N) : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int off = carry, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            if (w < wave) off += wsum[w];
            tot += wsum[w];
        }
        if (p < P) offs[p] = off + inc - v;
        __syncthreads();
        if (tid == 0) carry += tot;
        __syncthreads();
    }
    if (tid == 0) { offs[P] = carry; *total = carry; }
}

__global__ __launch_bounds__(256) void refine_rowmap_kernel(const int32_t* __restrict__ n_matches, const int32_t* __restrict__ offs,
                                                            int N, int32_t* __restrict__ rowmap) {
    const int p = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
    const int n = clampi(n_matches[p], 0, N);
    if (r < n) rowmap[offs[p] + r] = p * N + r;
}

void launch_refine_rowmap(const int32_t* n_matches, int P, int N, int32_t* offs, int32_t* rowmap, int32_t* total,
                          hipStream_t st) {
    refine_offsets_kernel<<<1, 1024, 0, st>>>(n_matches, P, N, offs, total);
    refine_rowmap_kernel<<<dim3(ceil_div(N, 256), P), 256, 0, st>>>(n_matches, offs, N, rowmap);
}

// wave per live row, lane = one of the 64 logits (flattened index = 8*y + x)
__global__ __launch_bounds__(256) void refine_rows_kernel(const float* __restrict__ o, const int32_t* __restrict__ rowmap,
                                                          const int32_t* __restrict__ total, const float* __restrict__ kp0,
                                                          const float* __restrict__ kp1, const float* __restrict__ scale0,
                                                          const int64_t* __restrict__ idx0, const int64_t* __restrict__ idx1,
                                                          int N, float fine_conf, float* __restrict__ rows_tmp,
                                                          unsigned char* __restrict__ keep_tmp) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= *total) return;
    const float z = o[(size_t)g * 64 + lane] * 3.f;
    const float m = wave_max(z);
    const float e = expf(z - m);
    const float s = wave_sum(e);
    const float pr = e / s;
    const float conf = wave_max(pr);
    const float ox = wave_sum(pr * (float)((lane & 7) - 4));
    const float oy = wave_sum(pr * (float)((lane >> 3) - 4));
    if (lane == 0) {
        const int code = rowmap[g];
        const int p = code / N;
        const size_t i0 = (size_t)p * N + (size_t)idx0[code];
        const size_t i1 = (size_t)p * N + (size_t)idx1[code];
        const float sc = scale0[i0];
        rows_tmp[(size_t)g * 4 + 0] = kp0[i0 * 2 + 0] + ox * sc;
        rows_tmp[(size_t)g * 4 + 1] = kp0[i0 * 2 + 1] + oy * sc;
        rows_tmp[(size_t)g * 4 + 2] = kp1[i1 * 2 + 0];
        rows_tmp[(size_t)g * 4 + 3] = kp1[i1 * 2 + 1];
        keep_tmp[g] = conf > fine_conf ? 1 : 0;
    }
}

// grid (P), block 1024: ordered compaction of the kept rows of each pair
__global__ __launch_bounds__(1024) void refine_compact_kernel(const float* __restrict__ rows_tmp,
                                                              const unsigned char* __restrict__ keep_tmp,
                                                              const int32_t* __restrict__ offs, int N,
                                                              float* __restrict__ out, int32_t* __restrict__ n_out) {
    __shared__ int wsum[16];
    __shared__ int s_base;
    const int p = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int beg = offs[p], n = offs[p + 1] - beg;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int r = base + tid;
        const bool keep = r < n && keep_tmp[beg + r];
        const unsigned long long bal = __ballot(keep);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int off = s_base, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            if (w < wave) off += wsum[w];
            tot += wsum[w];
        }
        if (keep) {
            const float4 v = *reinterpret_cast<const float4*>(rows_tmp + (size_t)(beg + r) * 4);
            *reinterpret_cast<float4*>(out + ((size_t)p * N + off + before) * 4) = v;
        }
        __syncthreads();
        if (tid == 0) s_base += tot;
        __syncthreads();
    }
    if (tid == 0) n_out[p] = s_base;
}

void launch_refine_finish(const float* o, const int32_t* rowmap, const int32_t* offs, const int32_t* total,
                          const float* kp0, const float* kp1, const float* scale0, const int64_t* idx0,
                          const int64_t* idx1, int P, int N, float fine_conf, float* out, int32_t* n_out,
                          float* rows_tmp, unsigned char* keep_tmp, hipStream_t st) {
    refine_rows_kernel<<<ceil_div(P * N, 4), 256, 0, st>>>(o, rowmap, total, kp0, kp1, scale0, idx0, idx1, N, fine_conf,
                                                          rows_tmp, keep_tmp);
    refine_compact_kernel<<<P, 1024, 0, st>>>(rows_tmp, keep_tmp, offs, N, out, n_out);
}

}  // namespace xfh
