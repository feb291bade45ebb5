// Row-major linear layers on the f32 matrix cores + the two head epilogue kernels.
//
//   y[M][N] = relu?( x[M][K] . W[K][N] + b )        x rows contiguous in K ("channels-last")
//
// Used for every pointwise (1x1) layer that runs on channels-last data:
//   heatmap_head.0/.1            (modules/model.py:79-84)   x = feats  (B*h*w, 64)
//   keypoint_head.0..3           (modules/model.py:87-92)   x = 8x8 unfold of the gray image
//                                (model.py:113-120,152), built on the fly by the UNFOLD8 loader
//   fine_matcher                 (modules/model.py:97-111)  x = cat(desc0[idx0], desc1[idx1]),
//                                gathered on the fly by the GATHER2 loader (xfeat.py:308-316)
//
// MFMA orientation: A = x (i = row), B = W (j = col): lane holds col j = l&31 and 16 rows, so
// row-major stores are 32 consecutive floats per half-wave.  One workgroup (4 waves) computes
// 256 rows x NT cols (NT = 64 or 32); each wave 64 rows (2 row blocks) x NT.  K is walked in
// chunks of 32: x chunk in LDS as [row][33] (odd stride -> conflict-free column reads),
// W chunk as [32][NT]; the next chunk is prefetched into registers during the MFMAs.
#include "kernels.hpp"
#include "bx_split.hpp"
#include "linear_fx_body.hpp"

namespace xfh {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int NT, int LOADER>
__global__ __launch_bounds__(256) void linear_mfma_kernel(const float* __restrict__ wkn, const float* __restrict__ bias,
                                                          int N, int n_pad, int relu, LinSrc s, int M,
                                                          const int32_t* __restrict__ m_dev, float* __restrict__ y,
                                                          int ldy) {
    constexpr int KC = 32, NKC = K / KC, NBLK = NT / 32, RB = 2, ROWS = 256, XS = 33;
    constexpr int WV = (KC * NT / 4 + 255) / 256;
    __shared__ float Xl[ROWS * XS];
    __shared__ __attribute__((aligned(16))) float Wl[KC * NT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int row0 = blockIdx.x * ROWS, n0 = blockIdx.y * NT;
    int Mlive = M;
    if (m_dev) Mlive = min(M, *m_dev);
    if (row0 >= Mlive) return;

    // this thread stages 8 rows (rl = tid/8 + 32*i), always the same float4 column q = tid%8
    const int q = tid & 7;
    long src0[8];
    long src1[8];   // GATHER2: second half of K
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = row0 + (tid >> 3) + 32 * i;
        src0[i] = -1; src1[i] = -1;
        if (row < Mlive) {
            if (LOADER == LOAD_ROWMAJOR) {
                src0[i] = (long)row * s.ldx + 4 * q;
            } else if (LOADER == LOAD_UNFOLD8) {
                const int wc = s.W >> 3, hw = (s.H >> 3) * wc;
                const int b = row / hw, rem = row - b * hw;
                const int ci = rem / wc, cj = rem - ci * wc;
                // k = 8*dy+dx ; chunk c covers dy = 4c + q/2, dx = 4*(q&1)..+3
                src0[i] = (long)b * s.H * s.W + (long)(8 * ci + (q >> 1)) * s.W + 8 * cj + 4 * (q & 1);
            } else {
                const int code = s.rowmap[row];
                const int p = code / s.N;
                src0[i] = ((long)p * s.N + (long)s.idx0[code]) * 64 + 4 * q;
                src1[i] = ((long)p * s.N + (long)s.idx1[code]) * 64 + 4 * q;
            }
        }
    }

    f32x16 acc[RB][NBLK];      // start at the bias (loaded before any store so the loads batch)
#pragma unroll
    for (int c = 0; c < NBLK; ++c) {
        const float bs = bias[n0 + c * 32 + l31];           // bias is padded to n_pad
#pragma unroll
        for (int a = 0; a < RB; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = bs;
    }

    float4 xreg[8];
    float4 wreg[WV];
    auto prefetch = [&](int kc) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src0[i] >= 0) {
                if (LOADER == LOAD_ROWMAJOR) v = *reinterpret_cast<const float4*>(s.x + src0[i] + kc * KC);
                else if (LOADER == LOAD_UNFOLD8) v = *reinterpret_cast<const float4*>(s.x + src0[i] + (long)(4 * kc) * s.W);
                else v = kc < 2 ? *reinterpret_cast<const float4*>(s.x + src0[i] + kc * KC)
                                : *reinterpret_cast<const float4*>(s.x2 + src1[i] + (kc - 2) * KC);
            }
            xreg[i] = v;
        }
#pragma unroll
        for (int v = 0; v < WV; ++v) {
            const int e = tid + v * 256;          // float4 index inside the [32][NT] chunk
            if (e < KC * NT / 4) {
                const int kr = e / (NT / 4), c4 = e - kr * (NT / 4);
                wreg[v] = *reinterpret_cast<const float4*>(wkn + (size_t)(kc * KC + kr) * n_pad + n0 + 4 * c4);
            }
        }
    };

    prefetch(0);
    for (int kc = 0; kc < NKC; ++kc) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float* d = Xl + ((tid >> 3) + 32 * i) * XS + 4 * q;
            d[0] = xreg[i].x; d[1] = xreg[i].y; d[2] = xreg[i].z; d[3] = xreg[i].w;
        }
#pragma unroll
        for (int v = 0; v < WV; ++v) {
            const int e = tid + v * 256;
            if (e < KC * NT / 4) reinterpret_cast<float4*>(Wl)[e] = wreg[v];
        }
        __syncthreads();
        if (kc + 1 < NKC) prefetch(kc + 1);
#pragma unroll
        for (int p = 0; p < KC / 2; ++p) {
            float a[RB], bb[NBLK];
#pragma unroll
            for (int r = 0; r < RB; ++r) a[r] = Xl[(wave * 64 + r * 32 + l31) * XS + 2 * p + half];
#pragma unroll
            for (int c = 0; c < NBLK; ++c) bb[c] = Wl[(2 * p + half) * NT + c * 32 + l31];
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int c = 0; c < NBLK; ++c)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], bb[c], acc[r][c], 0, 0, 0);
        }
    }

#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + wave * 64 + rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row < Mlive) {
#pragma unroll
                for (int c = 0; c < NBLK; ++c) {
                    const int col = n0 + c * 32 + l31;
                    if (col < N) {
                        float v = acc[rb][c][r];
                        if (relu) v = fmaxf(v, 0.f);
                        y[(size_t)row * ldy + col] = v;
                    }
                }
            }
        }
}

int launch_linear_mfma(const float* w_kn, const float* bias, int K, int N, int n_pad, bool relu, LinLoader loader,
                       const LinSrc& src, int M, const int32_t* m_dev, float* y, int ldy, hipStream_t st) {
    if (M <= 0) return 0;
    const int gx = ceil_div(M, 256);
    const int r = relu ? 1 : 0;
#define XFH_LIN(KV, NTV, LD)                                                                      \
    linear_mfma_kernel<KV, NTV, LD><<<dim3(gx, ceil_div(n_pad, NTV)), 256, 0, st>>>(w_kn, bias, N, n_pad, r, src, \
                                                                                   M, m_dev, y, ldy)
    const bool nt64 = (n_pad % 64 == 0);
    if (K == 64 && loader == LOAD_ROWMAJOR && nt64) { XFH_LIN(64, 64, LOAD_ROWMAJOR); return 0; }
    if (K == 64 && loader == LOAD_ROWMAJOR && !nt64) { XFH_LIN(64, 32, LOAD_ROWMAJOR); return 0; }
    if (K == 64 && loader == LOAD_UNFOLD8 && nt64) { XFH_LIN(64, 64, LOAD_UNFOLD8); return 0; }
    if (K == 96 && loader == LOAD_ROWMAJOR && nt64) { XFH_LIN(96, 64, LOAD_ROWMAJOR); return 0; }       // LighterGlue similarity matrix (d = 96)
    if (K == 128 && loader == LOAD_GATHER2 && nt64) { XFH_LIN(128, 64, LOAD_GATHER2); return 0; }
    if (K == 128 && loader == LOAD_ROWMAJOR && nt64) { XFH_LIN(128, 64, LOAD_ROWMAJOR); return 0; }
    if (K == 512 && loader == LOAD_ROWMAJOR && nt64) { XFH_LIN(512, 64, LOAD_ROWMAJOR); return 0; }
#undef XFH_LIN
    return -1;
}

// ------------------------------------------------------------------------------------------
// The same layer in the fp16-pair arithmetic (linear_fx_body.hpp; round 5): the fine_matcher's chain, activations in the split form between its layers.
template <int K, int IN, int OUT>
__global__ __launch_bounds__(256, 2) void linear_fx_kernel(LinFxArgs a, int cold) {
    kernel_entry_hooks(cold);      // debug: code-position shift / cold instruction cache (common.hpp)
    linear_fx_body<K, IN, OUT>(a);
}

template <int K>
__global__ __launch_bounds__(512, 2) void linear_fxd_kernel(LinFxArgs a, int cold) {
    kernel_entry_hooks(cold);
    linear_fxd_body<K>(a);
}

int launch_linear_fx(const void* w_fx, const float* bias, int K, int N, int n_pad, bool relu, LinLoader loader, const LinSrc& src, int M, const int32_t* m_dev,
                     void* y, int ldy, hipStream_t st, int* status, bool in_pair, bool out_pair) {
    if (M <= 0) return 0;
    if (!w_fx || n_pad % 64) return -1;
    if (out_pair ? (N != n_pad || ldy % 8) : (N % 4 || ldy % 4)) return -1;      // (the epilogues store 16 bytes at a time)
    if (in_pair && (loader != LOAD_ROWMAJOR || src.ldx % 8)) return -1;
    LinFxArgs a{};
    a.wq = reinterpret_cast<const uint4*>(w_fx); a.bias = bias; a.N = N; a.relu = relu ? 1 : 0;
    a.x = src.x; a.ldx = src.ldx; a.x2 = src.x2; a.idx0 = src.idx0; a.idx1 = src.idx1; a.rowmap = src.rowmap; a.cap = src.N;
    a.M = M; a.m_dev = m_dev; a.y = y; a.ldy = ldy; a.status = status;
    a.n_row_blocks = (ceil_div(M, 256) + 7) / 8 * 8;      // row blocks padded to a multiple of 8: one XCD per row block (xcd_group_map)
    a.n_col_blocks = n_pad / 64;
    const dim3 g(xcd_grid_size(a.n_col_blocks, a.n_row_blocks));
#define XFH_LFX(KV, INV, OUTV)                                                                                              \
    do {                                                                                                                    \
        static AttrMask done{0};                                                                                            \
        set_max_dynamic_lds(reinterpret_cast<const void*>(&linear_fx_kernel<KV, INV, OUTV>), linfx::LDS_BYTES, done);       \
        linear_fx_kernel<KV, INV, OUTV><<<g, 256, linfx::LDS_BYTES, st>>>(a, g_debug_cold);                                 \
        return 0;                                                                                                           \
    } while (0)
    if (K == 128 && !in_pair && out_pair && loader == LOAD_GATHER2) XFH_LFX(128, LFX_IN_GATHER2, LFX_OUT_PAIR);
    if (K == 128 && !in_pair && out_pair && loader == LOAD_ROWMAJOR) XFH_LFX(128, LFX_IN_F32, LFX_OUT_PAIR);
    if (K == 512 && in_pair && out_pair && n_pad % 128 == 0) {      // the chain's inner layers: 256 x 128 tiles, every operand by LDS-DMA
        a.n_col_blocks = n_pad / 128;
        static AttrMask done{0};
        set_max_dynamic_lds(reinterpret_cast<const void*>(&linear_fxd_kernel<512>), linfxd::LDS_BYTES, done);
        linear_fxd_kernel<512><<<dim3(xcd_grid_size(a.n_col_blocks, a.n_row_blocks)), linfxd::THREADS, linfxd::LDS_BYTES, st>>>(a, g_debug_cold);
        return 0;
    }
    if (K == 512 && in_pair && !out_pair) XFH_LFX(512, LFX_IN_PAIR, LFX_OUT_F32);
#undef XFH_LFX
    return -1;
}

// ------------------------------------------------------------------------------------------
// reliability = sigmoid(x . w + b), x (M,64) row-major: 16 lanes per row, float4 each
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dot_sigmoid_kernel(const float* __restrict__ x, int M, const float* __restrict__ w,
                                                          const float* __restrict__ b, float* __restrict__ out) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int row = g >> 4, part = g & 15;
    float s = 0.f;
    if (row < M) {
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)row * 64 + part * 4);
        const float4 ww = *reinterpret_cast<const float4*>(w + part * 4);
        s = v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
    }
    s += __shfl_xor(s, 8, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 1, 64);
    if (row < M && part == 0) out[row] = 1.f / (1.f + expf(-(s + b[0])));
}
void launch_dot_sigmoid(const float* x, int M, const float* w, const float* b, float* out, hipStream_t st) {
    dot_sigmoid_kernel<<<ceil_div(M * 16, 256), 256, 0, st>>>(x, M, w, b, out);
}

// ------------------------------------------------------------------------------------------
// heat[b][8i+dy][8j+dx] = softmax_65(logits[b,i,j,:])[8dy+dx]          (xfeat.py:242-247)
// one thread per cell; logits (B*hc*wc, 65) row-major
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_heat_kernel(const float* __restrict__ logits, int B, int hc, int wc,
                                                           float* __restrict__ heat) {
    const int cell = blockIdx.x * 256 + threadIdx.x;
    const int ncell = B * hc * wc;
    if (cell >= ncell) return;
    const float* p = logits + (size_t)cell * 65;
    float v[65];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 65; ++k) { v[k] = p[k]; mx = fmaxf(mx, v[k]); }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 65; ++k) { v[k] = expf(v[k] - mx); sum += v[k]; }
    const int hw = hc * wc;
    const int b = cell / hw, rem = cell - b * hw;
    const int i = rem / wc, j = rem - i * wc;
    const int W = wc * 8;
    float* o = heat + ((size_t)b * hc * 8 + 8 * i) * W + 8 * j;
#pragma unroll
    for (int dy = 0; dy < 8; ++dy) {
        float4 a = make_float4(v[8 * dy + 0] / sum, v[8 * dy + 1] / sum, v[8 * dy + 2] / sum, v[8 * dy + 3] / sum);
        float4 c = make_float4(v[8 * dy + 4] / sum, v[8 * dy + 5] / sum, v[8 * dy + 6] / sum, v[8 * dy + 7] / sum);
        *reinterpret_cast<float4*>(o + (size_t)dy * W) = a;
        *reinterpret_cast<float4*>(o + (size_t)dy * W + 4) = c;
    }
}
void launch_softmax_heat(const float* logits, int B, int hc, int wc, float* heat, hipStream_t st) {
    softmax_heat_kernel<<<ceil_div(B * hc * wc, 256), 256, 0, st>>>(logits, B, hc, wc, heat);
}

}  // namespace xfh
