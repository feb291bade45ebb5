// Stand-alone sparse sampler behind the reference's public `InterpolateSparse2d` module
//   modules/interpolator.py:10-33  (XFeat.interpolator, modules/xfeat.py:37)
// out[b,n,c] = grid_sample(x, normgrid(pos), mode, align_corners=False, zeros padding)[b,c,n]
// with the reference's fp32 operation order: g = 2*(p/(S-1)) - 1, u = fma(g+1, Sm/2, -0.5);
// nearest = round-half-even, bicubic = Keys A=-0.75 with per-tap zero padding.  The hot path does not come
// through here (its three sampling sites are fused into score_keys / descriptor kernels, k_detect.hip); this is the
// general (any C, any map size, NCHW) form for callers that use the module directly.
#include "kernels.hpp"

namespace xfh {

__device__ inline float sample_coord_f(float p, int S, int Sm) {
    const float q = p / (float)(S - 1);
    const float g = 2.0f * q - 1.0f;
    const float g1 = g + 1.0f;
    return __fmaf_rn(g1, (float)Sm * 0.5f, -0.5f);
}
__device__ inline float tap0(const float* __restrict__ m, int Hm, int Wm, int y, int x) {
    return (x >= 0 && x < Wm && y >= 0 && y < Hm) ? m[(size_t)y * Wm + x] : 0.f;
}
__device__ inline void cubic_taps(float t, float w[4]) {
    const float A = -0.75f;
    float x = t + 1.f;
    w[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
    x = t;
    w[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    x = 1.f - t;
    w[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    x = 2.f - t;
    w[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}

// thread = (b, n, c), c fastest: the C channel planes of one point are read by neighbouring lanes
__global__ __launch_bounds__(256) void sample_sparse_kernel(const float* __restrict__ x, const float* __restrict__ pos, int B, int C, int Hm,
                                                            int Wm, int N, int H, int W, int mode, float* __restrict__ out) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (size_t)B * N * C) return;
    const int c = (int)(g % C);
    const size_t bn = g / C;
    const int b = (int)(bn / N);
    const float px = pos[bn * 2 + 0], py = pos[bn * 2 + 1];
    const float ux = sample_coord_f(px, W, Wm), uy = sample_coord_f(py, H, Hm);
    const float* m = x + ((size_t)b * C + c) * Hm * Wm;
    float v;
    if (mode == 0) {                                     // nearest
        v = tap0(m, Hm, Wm, (int)rintf(uy), (int)rintf(ux));
    } else if (mode == 1) {                              // bilinear
        const float fx = floorf(ux), fy = floorf(uy);
        const float tx = ux - fx, ty = uy - fy;
        const int x0 = (int)fx, y0 = (int)fy;
        v = tap0(m, Hm, Wm, y0, x0) * ((1.f - tx) * (1.f - ty)) + tap0(m, Hm, Wm, y0, x0 + 1) * (tx * (1.f - ty)) +
            tap0(m, Hm, Wm, y0 + 1, x0) * ((1.f - tx) * ty) + tap0(m, Hm, Wm, y0 + 1, x0 + 1) * (tx * ty);
    } else {                                             // bicubic
        const float fx = floorf(ux), fy = floorf(uy);
        float wx[4], wy[4];
        cubic_taps(ux - fx, wx);
        cubic_taps(uy - fy, wy);
        const int x0 = (int)fx - 1, y0 = (int)fy - 1;
        v = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float row = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) row += tap0(m, Hm, Wm, y0 + r, x0 + i) * wx[i];
            v += row * wy[r];
        }
    }
    out[g] = v;
}

void launch_sample_sparse(const float* x, const float* pos, int B, int C, int Hm, int Wm, int N, int H, int W, int mode, float* out,
                          hipStream_t st) {
    const size_t total = (size_t)B * N * C;
    sample_sparse_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(x, pos, B, C, Hm, Wm, N, H, W, mode, out);
}

}  // namespace xfh
