// Pre-processing kernels: channel-mean + InstanceNorm2d(1), bilinear resize, pyramid sum.
// All HBM-bound streaming kernels: float4 rows, 64-lane wave reductions, fp64 statistics.
#include "kernels.hpp"

namespace xfh {

// ------------------------------------------------------------------------------------------
// gray = mean over C ; per-image mean / biased variance in fp64     (model.py:135-136)
// grid (GS_CHUNKS, B), block 256.  part[b][chunk][2] = {sum, sum of squares}
// ------------------------------------------------------------------------------------------
// CT: channel count known at compile time (1 or 3; 0 = any).  A thread's loads of GS_UNROLL rows x CT channels are issued before any
// of them is used (the runtime channel loop waited for every load on its own: one float4 in flight per thread, 4.4 TB/s at full
// occupancy); the arithmetic and its order are unchanged (channel sum c = 0, 1, ..., one division; fp64 sums in row order).
constexpr int GS_UNROLL = 5;          // VGA: 1200 float4 per chunk = 4.7 rows of 256 threads
template <int CT>
__global__ __launch_bounds__(256) void gray_stats_kernel(const float* __restrict__ img, int C, int HW,
                                                         double* __restrict__ part, float* __restrict__ gray) {
    const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
    const int n4 = HW >> 2;
    const int per = ceil_div(n4, GS_CHUNKS);
    const int beg = ch * per, end = min(beg + per, n4);
    const float4* base = reinterpret_cast<const float4*>(img + (size_t)b * C * HW);
    float4* gout = reinterpret_cast<float4*>(gray + (size_t)b * HW);
    const float fC = (float)C;
    double s = 0.0, q = 0.0;
    auto finish = [&](float4 a, int i) {
        a.x /= fC; a.y /= fC; a.z /= fC; a.w /= fC;
        gout[i] = a;      // raw channel mean; consumers apply the per-image (alpha, beta)
        s += (double)a.x + (double)a.y + (double)a.z + (double)a.w;
        q += (double)a.x * a.x + (double)a.y * a.y + (double)a.z * a.z + (double)a.w * a.w;
    };
    if constexpr (CT > 0) {
        for (int i0 = beg + tid; i0 < end; i0 += 256 * GS_UNROLL) {
            float4 v[GS_UNROLL][CT];
#pragma unroll
            for (int u = 0; u < GS_UNROLL; ++u) {
                const int i = min(i0 + 256 * u, end - 1);          // (rows beyond the chunk re-read its last element and are dropped below)
#pragma unroll
                for (int c = 0; c < CT; ++c) v[u][c] = base[(size_t)c * n4 + i];
            }
#pragma unroll
            for (int u = 0; u < GS_UNROLL; ++u) {
                float4 a = v[u][0];
#pragma unroll
                for (int c = 1; c < CT; ++c) { a.x += v[u][c].x; a.y += v[u][c].y; a.z += v[u][c].z; a.w += v[u][c].w; }
                if (i0 + 256 * u < end) finish(a, i0 + 256 * u);
            }
        }
    } else {
        for (int i = beg + tid; i < end; i += 256) {
            float4 a = base[i];
            for (int c = 1; c < C; ++c) {
                float4 v = base[(size_t)c * n4 + i];
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            finish(a, i);
        }
    }
    s = wave_sum(s);
    q = wave_sum(q);
    __shared__ double sm[8];
    if ((tid & 63) == 0) { sm[(tid >> 6) * 2] = s; sm[(tid >> 6) * 2 + 1] = q; }
    __syncthreads();
    if (tid == 0) {
        part[((size_t)b * GS_CHUNKS + ch) * 2 + 0] = sm[0] + sm[2] + sm[4] + sm[6];
        part[((size_t)b * GS_CHUNKS + ch) * 2 + 1] = sm[1] + sm[3] + sm[5] + sm[7];
    }
}

// uint8 ingest (XFeat.parse_input / .float(): modules/xfeat.py:396-403, 232): the same statistics + raw gray from
// (B,C,H,W) or (B,H,W,C) bytes.  Per channel v = float(u8) / divisor (IEEE division, divisor 255 for parse_input's
// numpy path, 1 for a uint8 tensor handed to detectAndCompute), then the channel mean exactly as the fp32 kernel does
// (sequential sum, one division) -- bit-identical to converting on the host, with a quarter of the bytes moved.
template <bool NHWC>
__global__ __launch_bounds__(256) void gray_stats_u8_kernel(const unsigned char* __restrict__ img, int C, int HW, float divisor,
                                                            double* __restrict__ part, float* __restrict__ gray) {
    const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
    const int n4 = HW >> 2;
    const int per = ceil_div(n4, GS_CHUNKS);
    const int beg = ch * per, end = min(beg + per, n4);
    const unsigned char* base = img + (size_t)b * C * HW;
    const float fC = (float)C;
    double s = 0.0, q = 0.0;
    for (int i = beg + tid; i < end; i += 256) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        if (NHWC) {
            const unsigned char* p = base + (size_t)i * 4 * C;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float acc = (float)p[k * C] / divisor;
                for (int c = 1; c < C; ++c) acc += (float)p[k * C + c] / divisor;
                a[k] = acc;
            }
        } else {
            uchar4 v = reinterpret_cast<const uchar4*>(base)[i];
            a[0] = (float)v.x / divisor; a[1] = (float)v.y / divisor; a[2] = (float)v.z / divisor; a[3] = (float)v.w / divisor;
            for (int c = 1; c < C; ++c) {
                v = reinterpret_cast<const uchar4*>(base + (size_t)c * HW)[i];
                a[0] += (float)v.x / divisor; a[1] += (float)v.y / divisor; a[2] += (float)v.z / divisor; a[3] += (float)v.w / divisor;
            }
        }
        float4 g4 = make_float4(a[0] / fC, a[1] / fC, a[2] / fC, a[3] / fC);
        reinterpret_cast<float4*>(gray + (size_t)b * HW)[i] = g4;
        s += (double)g4.x + (double)g4.y + (double)g4.z + (double)g4.w;
        q += (double)g4.x * g4.x + (double)g4.y * g4.y + (double)g4.z * g4.z + (double)g4.w * g4.w;
    }
    s = wave_sum(s);
    q = wave_sum(q);
    __shared__ double sm[8];
    if ((tid & 63) == 0) { sm[(tid >> 6) * 2] = s; sm[(tid >> 6) * 2 + 1] = q; }
    __syncthreads();
    if (tid == 0) {
        part[((size_t)b * GS_CHUNKS + ch) * 2 + 0] = sm[0] + sm[2] + sm[4] + sm[6];
        part[((size_t)b * GS_CHUNKS + ch) * 2 + 1] = sm[1] + sm[3] + sm[5] + sm[7];
    }
}

// InstanceNorm2d(1) as a per-image affine map: x = fmaf(gray, alpha, beta), alpha = 1/sqrt(var+eps), beta = -mean*alpha.
// The map is applied by the two consumers of the image (block1 tile load, key-point head operand load) instead of
// a separate pass over the image (one read + one write of the RGB / gray planes less per frame).
// grid (B), block 64.   coef[b] = {alpha, beta}
__global__ __launch_bounds__(64) void gray_coef_kernel(const double* __restrict__ part, int HW, float eps, float* __restrict__ coef) {
    const int b = blockIdx.x, tid = threadIdx.x;
    double s = part[((size_t)b * GS_CHUNKS + tid) * 2 + 0];
    double q = part[((size_t)b * GS_CHUNKS + tid) * 2 + 1];
    s = wave_sum(s);
    q = wave_sum(q);
    if (tid == 0) {
        const double mean = s / HW;
        double var = q / HW - mean * mean;
        if (var < 0) var = 0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        coef[2 * b] = invstd;
        coef[2 * b + 1] = -(float)mean * invstd;
    }
}

void launch_gray_norm(const float* img, int B, int C, int H, int W, double* part, float* gray, float* coef, hipStream_t st) {
    static_assert(GS_CHUNKS == 64, "gray_coef_kernel reduces one wave of partial sums");
    const int HW = H * W;
    if (C == 3) gray_stats_kernel<3><<<dim3(GS_CHUNKS, B), 256, 0, st>>>(img, C, HW, part, gray);
    else if (C == 1) gray_stats_kernel<1><<<dim3(GS_CHUNKS, B), 256, 0, st>>>(img, C, HW, part, gray);
    else gray_stats_kernel<0><<<dim3(GS_CHUNKS, B), 256, 0, st>>>(img, C, HW, part, gray);
    gray_coef_kernel<<<B, 64, 0, st>>>(part, HW, 1e-5f, coef);
}

void launch_gray_norm_u8(const unsigned char* img, bool nhwc, float divisor, int B, int C, int H, int W, double* part, float* gray,
                         float* coef, hipStream_t st) {
    const int HW = H * W;
    if (nhwc) gray_stats_u8_kernel<true><<<dim3(GS_CHUNKS, B), 256, 0, st>>>(img, C, HW, divisor, part, gray);
    else gray_stats_u8_kernel<false><<<dim3(GS_CHUNKS, B), 256, 0, st>>>(img, C, HW, divisor, part, gray);
    gray_coef_kernel<<<B, 64, 0, st>>>(part, HW, 1e-5f, coef);
}

// ------------------------------------------------------------------------------------------
// bilinear, align_corners=False (ATen area_pixel_compute_source_index, cubic=false):
//   src = max(scale*(dst+0.5)-0.5, 0) ; i0=(int)src ; i1=min(i0+1,in-1) ; l1=src-i0 ; l0=1-l1
//   out = wy0*(wx0*v00+wx1*v01) + wy1*(wx0*v10+wx1*v11)
// ------------------------------------------------------------------------------------------
__device__ inline void lin_coef(float scale, int dst, int in, int& i0, int& i1, float& l0, float& l1) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > in - 1) i0 = in - 1;
    i1 = min(i0 + 1, in - 1);
    l1 = fminf(fmaxf(s - (float)i0, 0.f), 1.f);
    l0 = 1.f - l1;
}
__device__ inline float bilerp(const float* __restrict__ p, int Win, int y0, int y1, int x0, int x1,
                               float wy0, float wy1, float wx0, float wx1) {
    const float v00 = p[y0 * Win + x0], v01 = p[y0 * Win + x1];
    const float v10 = p[y1 * Win + x0], v11 = p[y1 * Win + x1];
    return wy0 * (wx0 * v00 + wx1 * v01) + wy1 * (wx0 * v10 + wx1 * v11);
}
// the same expression on values a caller has fetched itself (pyramid53_kernel's exact-x2 / x4 path: one row window per float4 instead of four taps per pixel)
__device__ inline float bilerp_vals(float v00, float v01, float v10, float v11, float wy0, float wy1, float wx0, float wx1) {
    return wy0 * (wx0 * v00 + wx1 * v01) + wy1 * (wx0 * v10 + wx1 * v11);
}

__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ src, int Hin, int Win,
                                                              float* __restrict__ dst, int Hout, int Wout,
                                                              float sh, float sw) {
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int pl = blockIdx.z;
    if (ox >= Wout || oy >= Hout) return;
    int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
    lin_coef(sh, oy, Hin, y0, y1, wy0, wy1);
    lin_coef(sw, ox, Win, x0, x1, wx0, wx1);
    const float* p = src + (size_t)pl * Hin * Win;
    dst[((size_t)pl * Hout + oy) * Wout + ox] = bilerp(p, Win, y0, y1, x0, x1, wy0, wy1, wx0, wx1);
}

void launch_resize_bilinear(const float* src, int planes, int Hin, int Win, float* dst, int Hout, int Wout,
                            float sh, float sw, hipStream_t st) {
    resize_bilinear_kernel<<<dim3(ceil_div(Wout, 64), ceil_div(Hout, 4), planes), 256, 0, st>>>(
        src, Hin, Win, dst, Hout, Wout, sh, sw);
}

// ------------------------------------------------------------------------------------------
// Two successive bilinear resizes + channel mean + statistics without materialising either resized image:
//   F.interpolate(x, scale_factor=s)  (modules/xfeat.py:379-381)  ->  (Hm,Wm)
//   preprocess_tensor's resize to multiples of 32 (modules/xfeat.py:234-238)  ->  (Ho,Wo)
//   x.mean(dim=1) + InstanceNorm statistics (modules/model.py:135-136)
// The dual-scale dense path (1024^2 -> 614^2 -> 608^2 and -> 1331^2 -> 1312^2) otherwise writes and re-reads ~4 GB per
// 32-image batch.  Per 64x16 output tile and channel: stage 1 (input -> intermediate grid) lands in LDS rounded to fp32
// exactly as the materialised image would be, stage 2 interpolates from LDS with the same expression as
// resize_bilinear_kernel, the channel sum runs in gray_stats_kernel's order: the gray plane is bit-identical to the
// three-kernel path.  grid (GS_CHUNKS, B): workgroup `ch` walks tiles ch, ch+64, ... and writes ONE partial sum, so
// gray_coef_kernel and the run-to-run determinism are unchanged.
// ------------------------------------------------------------------------------------------
constexpr int R2_TW = 64, R2_TH = 16, R2_RW = 2 * R2_TW + 2, R2_RH = 2 * R2_TH + 2;      // LDS region for stage-2 steps < 2
constexpr int R2_MAXC = 4;
__device__ inline float buffer_load_f32(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
__global__ __launch_bounds__(256) void resize2_gray_stats_kernel(const float* __restrict__ img, int C, int Hin, int Win, int Hm, int Wm,
                                                                 float s1h, float s1w, int Ho, int Wo, float s2h, float s2w,
                                                                 double* __restrict__ part, float* __restrict__ gray) {
    extern __shared__ float mid[];                                 // [C][rh * rw] of the current tile (rw-strided rows)
    const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
    const int tiles_x = ceil_div(Wo, R2_TW), tiles = tiles_x * ceil_div(Ho, R2_TH);
    const float fC = (float)C;
    const size_t plane = (size_t)Hin * Win;
    const float* base = img + (size_t)b * C * plane;
    // buffer resource of this image (C planes; < 2 GiB, checked by the host): 32-bit offsets, channel plane as scalar offset
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)((size_t)C * plane * 4), 0x00020000);
    const int lx = (tid & 15) * 4, ly = tid >> 4;                  // this thread: 4 consecutive output pixels of tile row ly
    double s = 0.0, q = 0.0;
    for (int t = ch; t < tiles; t += GS_CHUNKS) {
        const int oy0 = (t / tiles_x) * R2_TH, ox0 = (t % tiles_x) * R2_TW;
        const int oy1 = min(oy0 + R2_TH, Ho) - 1, ox1 = min(ox0 + R2_TW, Wo) - 1;
        int ym0, ym1, xm0, xm1, d0, d1; float f0, f1;
        lin_coef(s2h, oy0, Hm, ym0, d1, f0, f1);
        lin_coef(s2h, oy1, Hm, d0, ym1, f0, f1);
        lin_coef(s2w, ox0, Wm, xm0, d1, f0, f1);
        lin_coef(s2w, ox1, Wm, d0, xm1, f0, f1);
        const int rh = ym1 - ym0 + 1, rw = xm1 - xm0 + 1, rsz = rh * rw;      // <= R2_RH x R2_RW (host checks the steps)
        __syncthreads();                                            // the previous tile's readers are done
        // stage 1: input -> intermediate grid, all channels of one region pixel per thread (coefficients computed once)
        for (int e = tid; e < rsz; e += 256) {
            const int ry = e / rw, rx = e - ry * rw;
            int iy0, iy1, ix0, ix1; float vy0, vy1, vx0, vx1;
            lin_coef(s1h, ym0 + ry, Hin, iy0, iy1, vy0, vy1);
            lin_coef(s1w, xm0 + rx, Win, ix0, ix1, vx0, vx1);
            // the four tap offsets once per region pixel (32-bit, bytes); the channel plane goes into the scalar offset of
            // the buffer load instead of 64-bit pointer arithmetic per tap and channel
            const int o00 = (iy0 * Win + ix0) * 4, o01 = (iy0 * Win + ix1) * 4, o10 = (iy1 * Win + ix0) * 4, o11 = (iy1 * Win + ix1) * 4;
            for (int c = 0; c < C; ++c) {
                const int so = c * (int)(plane * 4);
                const float v00 = buffer_load_f32(rs, o00, so), v01 = buffer_load_f32(rs, o01, so);
                const float v10 = buffer_load_f32(rs, o10, so), v11 = buffer_load_f32(rs, o11, so);
                mid[c * rsz + e] = vy0 * (vx0 * v00 + vx1 * v01) + vy1 * (vx0 * v10 + vx1 * v11);
            }
        }
        __syncthreads();
        // stage 2: intermediate -> output grid from LDS, channel mean in gray_stats_kernel's order
        const int oy = oy0 + ly, ox = ox0 + lx;
        if (oy <= oy1 && ox <= ox1) {
            int y0, y1; float wy0, wy1;
            lin_coef(s2h, oy, Hm, y0, y1, wy0, wy1);
            y0 -= ym0; y1 -= ym0;
            float a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a[k] = 0.f;
                if (ox + k <= ox1) {
                    int x0, x1; float wx0, wx1;
                    lin_coef(s2w, ox + k, Wm, x0, x1, wx0, wx1);
                    x0 -= xm0; x1 -= xm0;
                    for (int c = 0; c < C; ++c) {
                        const float v = bilerp(mid + c * rsz, rw, y0, y1, x0, x1, wy0, wy1, wx0, wx1);
                        a[k] = c == 0 ? v : a[k] + v;
                    }
                }
            }
            float* g = gray + ((size_t)b * Ho + oy) * Wo + ox;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (ox + k <= ox1) {
                    const float v = a[k] / fC;
                    g[k] = v;
                    s += (double)v;
                    q += (double)v * v;
                }
        }
    }
    s = wave_sum(s);
    q = wave_sum(q);
    __shared__ double sm[8];
    if ((tid & 63) == 0) { sm[(tid >> 6) * 2] = s; sm[(tid >> 6) * 2 + 1] = q; }
    __syncthreads();
    if (tid == 0) {
        part[((size_t)b * GS_CHUNKS + ch) * 2 + 0] = sm[0] + sm[2] + sm[4] + sm[6];
        part[((size_t)b * GS_CHUNKS + ch) * 2 + 1] = sm[1] + sm[3] + sm[5] + sm[7];
    }
}

// The same with the tile's INPUT region staged in LDS first (round 5; the dense workload's second kernel: 274 us per launch, waves 70 % at s_waitcnt --
// profiles/r05_pmc_dense_linear_fx.txt -- behind 12 four-byte gathers per intermediate pixel): the rows [iy_lo, iy_hi] x 16-byte-aligned columns that stage 1's taps can touch
// arrive as 16-byte loads, all of a thread's in flight at once and ONE TILE AHEAD (a twentieth of the requests at scale 1.3, a fifth at 0.6), and stage 1 takes its taps from LDS -- the same
// values through the same expression: the gray plane stays bit-identical.  Needs Win % 4 == 0 (16-byte-aligned rows); LDS: [mid: C x mid_cap][in: C x in_rows x in_w].
struct __attribute__((aligned(16))) R2Coef { int i0, i1; float l0, l1; };
constexpr int R2_NTAB = R2_RH + R2_RW + R2_TH + R2_TW;      // 244 entries <= 256 threads
static_assert(R2_NTAB <= 256, "one table entry per thread");
constexpr int R2_MAXL = 12;      // 16-byte requests of a thread per tile: request i = (channel i >> 2, rows r0 + 8 (i & 3)): C <= 3, in_rows <= 32 (48 registers: four waves per SIMD)
__global__ __launch_bounds__(256, 4) void resize2_gray_stats_lds_kernel(const float* __restrict__ img, int C, int Hin, int Win, int Hm, int Wm,
                                                                      float s1h, float s1w, int Ho, int Wo, float s2h, float s2w,
                                                                      double* __restrict__ part, float* __restrict__ gray, int mid_cap, int in_rows, int in_w) {
    extern __shared__ __attribute__((aligned(16))) float r2sm[];
    // the interpolation coefficients of a tile, ONE lin_coef per thread and tile instead of thirteen: [stage-1 rows R2_RH][stage-1 columns R2_RW][stage-2 rows][stage-2 columns]
    R2Coef* tab = reinterpret_cast<R2Coef*>(r2sm);
    float* mid = r2sm + 4 * R2_NTAB;                                // [C][rh * rw] of the current tile
    float* inr = mid + (size_t)C * mid_cap;                         // [C][in_rows][in_w] of the input (mid_cap is a multiple of 4: 16-byte aligned)
    const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;
    const int tiles_x = ceil_div(Wo, R2_TW), tiles = tiles_x * ceil_div(Ho, R2_TH);
    const float fC = (float)C;
    const size_t plane = (size_t)Hin * Win;
    const float* base = img + (size_t)b * C * plane;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)((size_t)C * plane * 4), 0x00020000);
    const int lx = (tid & 15) * 4, ly = tid >> 4;
    const int qx = tid & 31, r0 = tid >> 5;                        // stage 0: lane (piece qx of a row, rows r0, r0 + 8, ...): no division
    struct Tile { int oy0, ox0, oy1, ox1, ym0, xm0, rh, rw, iy_lo, ix_lo, nrow, nq; };
    auto tile_of = [&](int t) {
        Tile T;
        T.oy0 = (t / tiles_x) * R2_TH; T.ox0 = (t % tiles_x) * R2_TW;
        T.oy1 = min(T.oy0 + R2_TH, Ho) - 1; T.ox1 = min(T.ox0 + R2_TW, Wo) - 1;
        int ym1, xm1, d0, d1, iy_hi, ix_hi; float f0, f1;
        lin_coef(s2h, T.oy0, Hm, T.ym0, d1, f0, f1);
        lin_coef(s2h, T.oy1, Hm, d0, ym1, f0, f1);
        lin_coef(s2w, T.ox0, Wm, T.xm0, d1, f0, f1);
        lin_coef(s2w, T.ox1, Wm, d0, xm1, f0, f1);
        T.rh = ym1 - T.ym0 + 1; T.rw = xm1 - T.xm0 + 1;
        // the input rows / columns stage 1 can touch (lin_coef's indices are monotone in the destination index)
        lin_coef(s1h, T.ym0, Hin, T.iy_lo, d1, f0, f1);
        lin_coef(s1h, ym1, Hin, d0, iy_hi, f0, f1);
        lin_coef(s1w, T.xm0, Win, T.ix_lo, d1, f0, f1);
        lin_coef(s1w, xm1, Win, d0, ix_hi, f0, f1);
        T.ix_lo &= ~3;
        T.nrow = min(iy_hi - T.iy_lo + 1, in_rows); T.nq = min((ix_hi - T.ix_lo) / 4 + 1, in_w / 4);      // (the host sized the region for the steps: the clamps never bind)
        return T;
    };
    uint4 pre[R2_MAXL];
    // the requests of a tile's input region: all of a thread's in flight at once, and in flight while the PREVIOUS tile is computed
    auto request = [&](const Tile& T) {
#pragma unroll
        for (int i = 0; i < R2_MAXL; ++i) {
            const int c = i >> 2, r = r0 + 8 * (i & 3);
            pre[i] = make_uint4(0u, 0u, 0u, 0u);
            if (c < C && r < T.nrow && qx < T.nq)
                pre[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, ((T.iy_lo + r) * Win + T.ix_lo + 4 * qx) * 4, c * (int)(plane * 4), 0));
        }
    };
    double s = 0.0, q = 0.0;
    Tile T = tile_of(min(ch, tiles - 1));
    if (ch < tiles) request(T);
    for (int t = ch; t < tiles; t += GS_CHUNKS) {
        const int rsz = T.rh * T.rw;
        __syncthreads();                                            // the previous tile's readers are done
        // stage 0: the region lands in LDS
#pragma unroll
        for (int i = 0; i < R2_MAXL; ++i) {
            const int c = i >> 2, r = r0 + 8 * (i & 3);
            if (c < C && r < T.nrow && qx < T.nq) *reinterpret_cast<uint4*>(inr + (c * in_rows + r) * in_w + 4 * qx) = pre[i];
        }
        // this thread's entry of the coefficient tables (offsets already in the units the stages index with)
        {
            R2Coef cf{0, 0, 0.f, 0.f};
            if (tid < R2_RH) {
                if (tid < T.rh) { lin_coef(s1h, T.ym0 + tid, Hin, cf.i0, cf.i1, cf.l0, cf.l1); cf.i0 = min(cf.i0 - T.iy_lo, T.nrow - 1) * in_w; cf.i1 = min(cf.i1 - T.iy_lo, T.nrow - 1) * in_w; }      // (min: never beyond the staged rows, whatever the host's fp32 sizing rounded to)
            } else if (tid < R2_RH + R2_RW) {
                const int rx = tid - R2_RH;
                if (rx < T.rw) { lin_coef(s1w, T.xm0 + rx, Win, cf.i0, cf.i1, cf.l0, cf.l1); cf.i0 = min(cf.i0 - T.ix_lo, 4 * T.nq - 1); cf.i1 = min(cf.i1 - T.ix_lo, 4 * T.nq - 1); }
            } else if (tid < R2_RH + R2_RW + R2_TH) {
                const int oy = T.oy0 + tid - (R2_RH + R2_RW);
                if (oy <= T.oy1) { lin_coef(s2h, oy, Hm, cf.i0, cf.i1, cf.l0, cf.l1); cf.i0 = (cf.i0 - T.ym0) * T.rw; cf.i1 = (cf.i1 - T.ym0) * T.rw; }
            } else if (tid < R2_NTAB) {
                const int ox = T.ox0 + tid - (R2_RH + R2_RW + R2_TH);
                if (ox <= T.ox1) { lin_coef(s2w, ox, Wm, cf.i0, cf.i1, cf.l0, cf.l1); cf.i0 -= T.xm0; cf.i1 -= T.xm0; }
            }
            if (tid < R2_NTAB) tab[tid] = cf;
        }
        Tile Tn = T;
        if (t + GS_CHUNKS < tiles) { Tn = tile_of(t + GS_CHUNKS); request(Tn); }
        __syncthreads();
        const unsigned rw_magic = T.rw > 1 ? 0xffffffffu / (unsigned)T.rw + 1u : 0u;      // floor(e / rw) = umulhi(e, magic) for e < 2^16 (one division per tile instead of one per pixel)
        // stage 1: input -> intermediate grid from LDS, all channels of one region pixel per thread (coefficients computed once)
        for (int e = tid; e < rsz; e += 256) {
            const int ry = rw_magic ? (int)__umulhi((unsigned)e, rw_magic) : e, rx = e - ry * T.rw;      // (a one-column region: the magic number would overflow to 0)
            const R2Coef cy = tab[ry], cx = tab[R2_RH + rx];
            const float vy0 = cy.l0, vy1 = cy.l1, vx0 = cx.l0, vx1 = cx.l1;
            const int o00 = cy.i0 + cx.i0, o01 = cy.i0 + cx.i1, o10 = cy.i1 + cx.i0, o11 = cy.i1 + cx.i1;
            for (int c = 0; c < C; ++c) {
                const float* p = inr + (size_t)c * in_rows * in_w;
                const float v00 = p[o00], v01 = p[o01], v10 = p[o10], v11 = p[o11];
                mid[c * rsz + e] = vy0 * (vx0 * v00 + vx1 * v01) + vy1 * (vx0 * v10 + vx1 * v11);
            }
        }
        __syncthreads();
        // stage 2: intermediate -> output grid from LDS, channel mean in gray_stats_kernel's order
        const int oy = T.oy0 + ly, ox = T.ox0 + lx;
        if (oy <= T.oy1 && ox <= T.ox1) {
            const R2Coef cy = tab[R2_RH + R2_RW + ly];
            const float wy0 = cy.l0, wy1 = cy.l1;
            float a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a[k] = 0.f;
                if (ox + k <= T.ox1) {
                    const R2Coef cx = tab[R2_RH + R2_RW + R2_TH + lx + k];
                    const float wx0 = cx.l0, wx1 = cx.l1;
                    const int o00 = cy.i0 + cx.i0, o01 = cy.i0 + cx.i1, o10 = cy.i1 + cx.i0, o11 = cy.i1 + cx.i1;
                    for (int c = 0; c < C; ++c) {
                        const float* p = mid + c * rsz;      // (bilerp's expression)
                        const float v = wy0 * (wx0 * p[o00] + wx1 * p[o01]) + wy1 * (wx0 * p[o10] + wx1 * p[o11]);
                        a[k] = c == 0 ? v : a[k] + v;
                    }
                }
            }
            float* g = gray + ((size_t)b * Ho + oy) * Wo + ox;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (ox + k <= T.ox1) {
                    const float v = a[k] / fC;
                    g[k] = v;
                    s += (double)v;
                    q += (double)v * v;
                }
        }
        T = Tn;
    }
    s = wave_sum(s);
    q = wave_sum(q);
    __shared__ double sm2[8];
    if ((tid & 63) == 0) { sm2[(tid >> 6) * 2] = s; sm2[(tid >> 6) * 2 + 1] = q; }
    __syncthreads();
    if (tid == 0) {
        part[((size_t)b * GS_CHUNKS + ch) * 2 + 0] = sm2[0] + sm2[2] + sm2[4] + sm2[6];
        part[((size_t)b * GS_CHUNKS + ch) * 2 + 1] = sm2[1] + sm2[3] + sm2[5] + sm2[7];
    }
}

// returns -1 when the stage-2 step is too large for the LDS region (callers then materialise the images)
int launch_gray_norm_resized(const float* img, int B, int C, int Hin, int Win, int Hm, int Wm, float s1h, float s1w, int Ho, int Wo,
                             float s2h, float s2w, double* part, float* gray, float* coef, hipStream_t st, int form) {
    if (!(s2h > 0.f) || !(s2w > 0.f) || s2h * (R2_TH - 1) + 3.f > (float)R2_RH || s2w * (R2_TW - 1) + 3.f > (float)R2_RW || C > R2_MAXC || (size_t)C * Hin * Win * 4 >= (1ull << 31)) return -1;
    // LDS: C planes of the largest intermediate region this (s2h, s2w) can need
    const int rh = (int)(s2h * (R2_TH - 1)) + 3, rw = (int)(s2w * (R2_TW - 1)) + 3;
    const size_t lds = (size_t)C * rh * rw * sizeof(float);
    if (lds > 72 * 1024) return -1;
    // the input region of a tile: rows / columns the rh x rw intermediate pixels can touch, the columns widened to 16-byte pieces
    const int in_rows = (int)(s1h * (rh - 1)) + 3, in_w = ((int)(s1w * (rw - 1)) + 3 + 3 + 3) / 4 * 4, mid_cap = (rh * rw + 3) / 4 * 4;
    const size_t lds2 = ((size_t)4 * R2_NTAB + (size_t)C * mid_cap + (size_t)C * in_rows * in_w) * sizeof(float);
    if (form == 1 && Win % 4 == 0 && (reinterpret_cast<uintptr_t>(img) & 15) == 0 && s1h > 0.f && s1w > 0.f && in_w <= 128 && in_rows <= 32 && C <= 3 && lds2 <= 78 * 1024) {      // (two workgroups per CU)
        static AttrMask attr2 = 0;
        set_max_dynamic_lds(reinterpret_cast<const void*>(resize2_gray_stats_lds_kernel), 78 * 1024, attr2);
        resize2_gray_stats_lds_kernel<<<dim3(GS_CHUNKS, B), 256, lds2, st>>>(img, C, Hin, Win, Hm, Wm, s1h, s1w, Ho, Wo, s2h, s2w, part, gray, mid_cap, in_rows, in_w);
        gray_coef_kernel<<<B, 64, 0, st>>>(part, Ho * Wo, 1e-5f, coef);
        return 0;
    }
    static AttrMask attr = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(resize2_gray_stats_kernel), 72 * 1024, attr);
    resize2_gray_stats_kernel<<<dim3(GS_CHUNKS, B), 256, lds, st>>>(img, C, Hin, Win, Hm, Wm, s1h, s1w, Ho, Wo, s2h, s2w, part, gray);
    gray_coef_kernel<<<B, 64, 0, st>>>(part, Ho * Wo, 1e-5f, coef);
    return 0;
}

// out = x3 + up(x4 -> x3 size) + up(x5 -> x3 size)      (model.py:146-148), NCHW planes.
// One workgroup per plane: the small x4 / x5 source planes are staged in LDS (coalesced), the
// x3 / out streams are float4.  Falls back to direct gathers when the planes exceed LDS.
__device__ inline float bilerp_at(const float* __restrict__ p, int Hs, int Ws, float sy, float sx, int oy, int ox) {
    int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
    lin_coef(sy, oy, Hs, y0, y1, wy0, wy1);
    lin_coef(sx, ox, Ws, x0, x1, wx0, wx1);
    return bilerp(p, Ws, y0, y1, x0, x1, wy0, wy1, wx0, wx1);
}

struct __attribute__((aligned(16))) PyrCoef { int i0, i1; float l0, l1; };

template <bool USE_LDS>
__global__ __launch_bounds__(256) void pyramid_sum_kernel(const float* __restrict__ x3, const float* __restrict__ x4,
                                                          const float* __restrict__ x5, float* __restrict__ out,
                                                          int H3, int W3, int H4, int W4, int H5, int W5) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int pl = blockIdx.x, tid = threadIdx.x;
    const float* p4 = x4 + (size_t)pl * H4 * W4;
    const float* p5 = x5 + (size_t)pl * H5 * W5;
    const float s4y = (float)H4 / (float)H3, s4x = (float)W4 / (float)W3;
    const float s5y = (float)H5 / (float)H3, s5x = (float)W5 / (float)W3;
    const size_t base = (size_t)pl * H3 * W3;
    const int n = H3 * W3;
    // the first five float4 of x3 per thread (a whole 60x80 plane) are requested BEFORE the x4 / x5 planes are staged: one
    // round trip instead of a dozen in a row (rolled loops wait for every load; the kernel ran at 4.4 TB/s on occupancy alone)
    constexpr int NPRE = 5;
    float4 pre[NPRE];
    const bool vec = (W3 & 3) == 0;
    if (vec) {
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int e4 = tid + k * 256;
            pre[k] = e4 < n / 4 ? *reinterpret_cast<const float4*>(x3 + base + 4 * (size_t)e4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (USE_LDS) {
        constexpr int NS = 6;                                  // 6 x 256 floats in flight per pass
        for (int e0 = 0; e0 < H4 * W4 + H5 * W5; e0 += NS * 256) {
            float t[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int e = e0 + k * 256 + tid;
                t[k] = e < H4 * W4 ? p4[e] : (e < H4 * W4 + H5 * W5 ? p5[e - H4 * W4] : 0.f);
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int e = e0 + k * 256 + tid;
                if (e < H4 * W4 + H5 * W5) sm[e] = t[k];
            }
        }
        // interpolation coefficients of every output column / row, once per plane instead of once per pixel (the kernel was bound by
        // its vector ops -- ten lin_coef per float4 -- not by HBM): tab[level][W3 columns | H3 rows] = {i0, i1, l0, l1}, the values
        // lin_coef returns, so that bilerp sees the same operands as before
        PyrCoef* tab = reinterpret_cast<PyrCoef*>(sm + ((H4 * W4 + H5 * W5 + 3) & ~3));
        for (int e = tid; e < 2 * (W3 + H3); e += 256) {
            const int lv = e >= W3 + H3, r = e - lv * (W3 + H3);
            PyrCoef c;
            if (r < W3) lin_coef(lv ? s5x : s4x, r, lv ? W5 : W4, c.i0, c.i1, c.l0, c.l1);
            else lin_coef(lv ? s5y : s4y, r - W3, lv ? H5 : H4, c.i0, c.i1, c.l0, c.l1);
            tab[e] = c;
        }
        __syncthreads();
        p4 = sm;
        p5 = sm + H4 * W4;
        if (vec) {
            auto emit = [&](int e4, const float4 v) {
                const int e = e4 * 4, oy = e / W3, ox = e - oy * W3;
                const PyrCoef y4 = tab[W3 + oy], y5 = tab[W3 + H3 + W3 + oy];
                const float vin[4] = {v.x, v.y, v.z, v.w};
                float r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const PyrCoef c4 = tab[ox + k], c5 = tab[W3 + H3 + ox + k];
                    r[k] = (vin[k] + bilerp(p4, W4, y4.i0, y4.i1, c4.i0, c4.i1, y4.l0, y4.l1, c4.l0, c4.l1))
                         + bilerp(p5, W5, y5.i0, y5.i1, c5.i0, c5.i1, y5.l0, y5.l1, c5.l0, c5.l1);
                }
                *reinterpret_cast<float4*>(out + base + e) = make_float4(r[0], r[1], r[2], r[3]);
            };
#pragma unroll
            for (int k = 0; k < NPRE; ++k) {
                const int e4 = tid + k * 256;
                if (e4 < n / 4) emit(e4, pre[k]);
            }
            for (int e4 = tid + NPRE * 256; e4 < n / 4; e4 += 256) emit(e4, *reinterpret_cast<const float4*>(x3 + base + 4 * (size_t)e4));
            return;
        }
    }
    if (vec) {
        auto emit = [&](int e4, const float4 v) {
            const int e = e4 * 4, oy = e / W3, ox = e - oy * W3;
            float4 r;
            r.x = (v.x + bilerp_at(p4, H4, W4, s4y, s4x, oy, ox)) + bilerp_at(p5, H5, W5, s5y, s5x, oy, ox);
            r.y = (v.y + bilerp_at(p4, H4, W4, s4y, s4x, oy, ox + 1)) + bilerp_at(p5, H5, W5, s5y, s5x, oy, ox + 1);
            r.z = (v.z + bilerp_at(p4, H4, W4, s4y, s4x, oy, ox + 2)) + bilerp_at(p5, H5, W5, s5y, s5x, oy, ox + 2);
            r.w = (v.w + bilerp_at(p4, H4, W4, s4y, s4x, oy, ox + 3)) + bilerp_at(p5, H5, W5, s5y, s5x, oy, ox + 3);
            *reinterpret_cast<float4*>(out + base + e) = r;
        };
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int e4 = tid + k * 256;
            if (e4 < n / 4) emit(e4, pre[k]);
        }
        for (int e4 = tid + NPRE * 256; e4 < n / 4; e4 += 256) emit(e4, *reinterpret_cast<const float4*>(x3 + base + 4 * (size_t)e4));
    } else {
        for (int e = tid; e < n; e += 256) {
            const int oy = e / W3, ox = e - oy * W3;
            out[base + e] = (x3[base + e] + bilerp_at(p4, H4, W4, s4y, s4x, oy, ox)) + bilerp_at(p5, H5, W5, s5y, s5x, oy, ox);
        }
    }
}

// block5.3 (1x1 convolution 128 -> 64, BN folded, ReLU; modules/model.py:78) AND the pyramid sum (model.py:146-148) in one launch: x5 never reaches HBM and the 1x1 has no
// launch of its own (as a kernel of its own on the f32 matrix cores it took 17 us per 64-frame step for 0.3 GFLOP -- a launch, a weight prologue and 19 200 positions).
// One workgroup = CG consecutive output channels of one image: it computes ITS x5 planes from block5.2's 128-channel output straight into LDS, stages its x4 planes
// beside them and then runs pyramid_sum_kernel's emit arithmetic over its CG x3 planes (contiguous in memory).  Every x5 plane is computed exactly once; the only
// redundancy is that the 64 / CG workgroups of an image each read block5.2's output (153 KB at VGA, L2-resident).  The kernel is made of memory latencies, so every
// phase keeps many independent loads in flight:
//   1x1      K split over the four waves (wave w: input channels 32 w .. + 31, fp32 fma chain in ascending order; the CG x 128 weights by wave-uniform scalar loads from
//            the [k][cout] image, two couts per v_pk_fma_f32), a lane owning positions lane + 64 j: P53_PP x 8 loads in flight per lane; the four partial sums meet in LDS
//            (where the x4 planes go afterwards) and are added in the fixed order ((w0 + w1) + w2) + w3, then bias and ReLU;
//   x4       requested into registers behind the 1x1, written to LDS behind the reduction;
//   emit     batches of P53_NB float4 of x3, the next batch requested before the current one is interpolated.
// Needs W3 % 4 == 0 and the planes to fit in LDS (launch_pyramid53 checks).
constexpr int P53_PP = 5, P53_KB = 8, P53_NB = 6, P53_X4R = 20;
template <int CG>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))      // 128 registers: four workgroups per CU = the whole VGA batch (1024 workgroups) in ONE round
void pyramid53_kernel(const float* __restrict__ x3, const float* __restrict__ x4, const float* __restrict__ y5,
                                                        const float* __restrict__ w53 /* [128][64] */, const float* __restrict__ b53, int relu53,
                                                        float* __restrict__ out, int H3, int W3, int H4, int W4, int H5, int W5, int generic_taps) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    static_assert(64 % CG == 0 && CG % 2 == 0, "channel groups");
    constexpr int NG = 64 / CG;
    const int b = blockIdx.x / NG, c0 = (blockIdx.x - b * NG) * CG, tid = threadIdx.x;
    const int n3 = H3 * W3, n4 = H4 * W4, n5 = H5 * W5;
    const int n4s = max(CG * n4, 4 * CG * n5);      // the x4 planes' region also holds the 1x1's four partial sums before
    float* s4 = sm;                                 // [CG][n4]   (before: [wave 4][CG][n5])
    float* s5 = sm + ((n4s + 3) & ~3);              // [CG][n5]
    PyrCoef* tab = reinterpret_cast<PyrCoef*>(s5 + ((CG * n5 + 3) & ~3));
    float2* wx = reinterpret_cast<float2*>(tab + 2 * (W3 + H3));      // [level 2][W3] column weights {l0, l1} alone: four consecutive columns = two 16-byte reads
    // the backbone's maps are exact halves / quarters of each other: a float4 of outputs then needs the SAME four columns of an x4 row and three of an x5 row
    // whatever the pixel (the border cases fall out of clamped column indices: a tap the table weights with 0 may be fetched from the neighbouring column)
    const bool exact = !generic_taps && H3 == 2 * H4 && W3 == 2 * W4 && H3 == 4 * H5 && W3 == 4 * W5;
    const size_t base3 = ((size_t)b * 64 + c0) * n3;
    const int nq = CG * n3 / 4;                     // float4 of this workgroup's x3 / out planes (W3 % 4 == 0)
    const float* p4g = x4 + ((size_t)b * 64 + c0) * n4;
    // ---- block5.3, this wave's quarter of K: partial[wv][c][p] = sum_{k in quarter} w[k][c0 + c] y5[k][p]
    {
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63;      // (wave-uniform: the weights come by scalar loads)
        const float* wq = w53 + (size_t)(32 * wv) * 64 + c0;
        const float* yq = y5 + ((size_t)b * 128 + 32 * wv) * n5;
        for (int p0 = 0; p0 < n5; p0 += 64 * P53_PP) {
            float acc[P53_PP][CG];
            int po[P53_PP];
#pragma unroll
            for (int j = 0; j < P53_PP; ++j) {
                po[j] = min(p0 + ln + 64 * j, n5 - 1);          // (positions beyond the map: a copy of the last one, never stored)
#pragma unroll
                for (int c = 0; c < CG; ++c) acc[j][c] = 0.f;
            }
#pragma unroll 1
            for (int k0 = 0; k0 < 32; k0 += P53_KB) {      // (rolled: unrolled, hipcc requests all 160 values at once -- 229 registers, two waves per SIMD)
                float v[P53_KB][P53_PP];                        // P53_KB x P53_PP loads in flight, then their fmas (k ascending per position)
#pragma unroll
                for (int kk = 0; kk < P53_KB; ++kk)
#pragma unroll
                    for (int j = 0; j < P53_PP; ++j) v[kk][j] = yq[(size_t)(k0 + kk) * n5 + po[j]];
#pragma unroll
                for (int kk = 0; kk < P53_KB; ++kk)
#pragma unroll
                    for (int j = 0; j < P53_PP; ++j)
#pragma unroll
                        for (int c = 0; c < CG; ++c) acc[j][c] = fmaf(v[kk][j], wq[(k0 + kk) * 64 + c], acc[j][c]);
            }
#pragma unroll
            for (int j = 0; j < P53_PP; ++j) {
                const int p = p0 + ln + 64 * j;
                if (p < n5) {
#pragma unroll
                    for (int c = 0; c < CG; ++c) s4[(wv * CG + c) * n5 + p] = acc[j][c];
                }
            }
        }
    }
    // the x4 planes (CG contiguous planes): requested now, written to LDS behind the reduction of the partial sums that occupy their place
    float r4[P53_X4R];
#pragma unroll
    for (int k = 0; k < P53_X4R; ++k) {
        const int e = tid + k * 256;
        r4[k] = p4g[min(e, CG * n4 - 1)];
    }
    // ... and the first batch of x3 (both travel under the coefficient tables, the barriers and the reduction)
    float4 cur[P53_NB];
#pragma unroll
    for (int k = 0; k < P53_NB; ++k) {
        const int e4 = tid + k * 256;
        cur[k] = *reinterpret_cast<const float4*>(x3 + base3 + 4 * (size_t)min(e4, nq - 1));      // (clamped, not branched: the requests leave back to back)
    }
    // interpolation coefficients of every output column / row (as pyramid_sum_kernel: the same operands, the same results)
    {
        const float s4y = (float)H4 / (float)H3, s4x = (float)W4 / (float)W3;
        const float s5y = (float)H5 / (float)H3, s5x = (float)W5 / (float)W3;
        for (int e = tid; e < 2 * (W3 + H3); e += 256) {
            const int lv = e >= W3 + H3, r = e - lv * (W3 + H3);
            PyrCoef c;
            if (r < W3) lin_coef(lv ? s5x : s4x, r, lv ? W5 : W4, c.i0, c.i1, c.l0, c.l1);
            else {
                lin_coef(lv ? s5y : s4y, r - W3, lv ? H5 : H4, c.i0, c.i1, c.l0, c.l1);
                if (exact) { c.i0 *= lv ? W5 : W4; c.i1 *= lv ? W5 : W4; }      // (the row-window form reads ROW OFFSETS here: one multiplication per table entry instead of four per float4)
            }
            tab[e] = c;
            if (r < W3) wx[lv * W3 + r] = make_float2(c.l0, c.l1);
        }
    }
    __syncthreads();
    {   // the four partial sums in a fixed order, bias, ReLU -> the x5 planes
        const float floor_y = relu53 ? 0.f : -INFINITY;
        for (int e = tid; e < CG * n5; e += 256) {
            const int c = e / n5;
            const float sum = ((s4[e] + s4[CG * n5 + e]) + s4[2 * CG * n5 + e]) + s4[3 * CG * n5 + e];
            s5[e] = fmaxf(sum + b53[c0 + c], floor_y);
        }
    }
    __syncthreads();
    // ---- the x4 planes take the partial sums' place
#pragma unroll
    for (int k = 0; k < P53_X4R; ++k) {
        const int e = tid + k * 256;
        if (e < CG * n4) s4[e] = r4[k];
    }
    for (int e = tid + P53_X4R * 256; e < CG * n4; e += 256) s4[e] = p4g[e];
    __syncthreads();
    auto emit = [&](int e4, const float4 v) {
        const int eg = e4 * 4, pl = eg / n3, e = eg - pl * n3, oy = e / W3, ox = e - oy * W3;
        const float* p4 = s4 + pl * n4;
        const float* p5 = s5 + pl * n5;
        const PyrCoef y4 = tab[W3 + oy], y5c = tab[W3 + H3 + W3 + oy];
        const float vin[4] = {v.x, v.y, v.z, v.w};
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const PyrCoef c4 = tab[ox + k], c5 = tab[W3 + H3 + ox + k];
            r[k] = (vin[k] + bilerp(p4, W4, y4.i0, y4.i1, c4.i0, c4.i1, y4.l0, y4.l1, c4.l0, c4.l1))
                 + bilerp(p5, W5, y5c.i0, y5c.i1, c5.i0, c5.i1, y5c.l0, y5c.l1, c5.l0, c5.l1);
        }
        *reinterpret_cast<float4*>(out + base3 + eg) = make_float4(r[0], r[1], r[2], r[3]);
    };
    // exact x2 / x4 maps: per float4 two row coefficients, four 16-byte reads of column weights, 3 + 3 reads per x4 / x5 row (the generic form: 10 + 32 reads)
    // (plane and row of a float4 by reciprocal multiplication: e4 < 2^18, launch_pyramid53 -- the quotient is off an integer by >= 1 / (2 W3), the product by < 1e-3 of that)
    const float inv_nq3 = 4.f / (float)n3, inv_w4 = 4.f / (float)W3;
    const int n3q = n3 >> 2, w3q = W3 >> 2;
    auto emit_exact = [&](int e4, const float4 v) {
        const int pl = (int)(((float)e4 + 0.5f) * inv_nq3), f4 = e4 - __mul24(pl, n3q);      // float4 index inside the plane  (24-bit multiplications: full rate, every factor is below 2^20)
        const int oy = (int)(((float)f4 + 0.5f) * inv_w4), q = f4 - __mul24(oy, w3q), ox = 4 * q, eg = 4 * e4;
        const PyrCoef y4 = tab[W3 + oy], y5c = tab[W3 + H3 + W3 + oy];
        const float4 wa = *reinterpret_cast<const float4*>(wx + ox), wb = *reinterpret_cast<const float4*>(wx + ox + 2);                  // x4: {l0, l1} of pixels 0, 1 | 2, 3
        const float4 wc = *reinterpret_cast<const float4*>(wx + W3 + ox), wd = *reinterpret_cast<const float4*>(wx + W3 + ox + 2);        // x5
        const float* p4 = s4 + __mul24(pl, n4);
        const float* r40 = p4 + y4.i0;             // (row offsets: see the table)
        const float* r41 = p4 + y4.i1;
        const int cl = max(2 * q - 1, 0), cr = min(2 * q + 2, W4 - 1);
        const float a00 = r40[cl], a03 = r40[cr], a10 = r41[cl], a13 = r41[cr];
        const float2 a0m = *reinterpret_cast<const float2*>(r40 + 2 * q), a1m = *reinterpret_cast<const float2*>(r41 + 2 * q);
        const float* p5 = s5 + __mul24(pl, n5);
        const float* r50 = p5 + y5c.i0;
        const float* r51 = p5 + y5c.i1;
        const int dl = max(q - 1, 0), dr = min(q + 1, W5 - 1);
        const float b00 = r50[dl], b01 = r50[q], b02 = r50[dr], b10 = r51[dl], b11 = r51[q], b12 = r51[dr];
        float4 r;
        r.x = (v.x + bilerp_vals(a00, a0m.x, a10, a1m.x, y4.l0, y4.l1, wa.x, wa.y)) + bilerp_vals(b00, b01, b10, b11, y5c.l0, y5c.l1, wc.x, wc.y);
        r.y = (v.y + bilerp_vals(a0m.x, a0m.y, a1m.x, a1m.y, y4.l0, y4.l1, wa.z, wa.w)) + bilerp_vals(b00, b01, b10, b11, y5c.l0, y5c.l1, wc.z, wc.w);
        r.z = (v.z + bilerp_vals(a0m.x, a0m.y, a1m.x, a1m.y, y4.l0, y4.l1, wb.x, wb.y)) + bilerp_vals(b01, b02, b11, b12, y5c.l0, y5c.l1, wd.x, wd.y);
        r.w = (v.w + bilerp_vals(a0m.y, a03, a1m.y, a13, y4.l0, y4.l1, wb.z, wb.w)) + bilerp_vals(b01, b02, b11, b12, y5c.l0, y5c.l1, wd.z, wd.w);
        *reinterpret_cast<float4*>(out + base3 + eg) = r;
    };
    for (int q0 = 0; q0 < nq; q0 += P53_NB * 256) {
        float4 nxt[P53_NB];
#pragma unroll
        for (int k = 0; k < P53_NB; ++k) {        // the next batch's requests travel under this batch's arithmetic
            const int e4 = q0 + P53_NB * 256 + tid + k * 256;
            nxt[k] = *reinterpret_cast<const float4*>(x3 + base3 + 4 * (size_t)min(e4, nq - 1));
        }
#pragma unroll
        for (int k = 0; k < P53_NB; ++k) {
            const int e4 = q0 + tid + k * 256;
            if (e4 < nq) { if (exact) emit_exact(e4, cur[k]); else emit(e4, cur[k]); }
        }
#pragma unroll
        for (int k = 0; k < P53_NB; ++k) cur[k] = nxt[k];
    }
}

constexpr int PYR53_CG = 4;
// -1: not this kernel's case (the caller runs block5.3 and launch_pyramid_sum)
int launch_pyramid53(const ConvW& c53, const float* x3, const float* x4, const float* y5, float* out, int B,
                     int H3, int W3, int H4, int W4, int H5, int W5, hipStream_t st) {
    if (c53.ks != 1 || c53.cin != 128 || c53.cout != 64 || c53.cout_pad != 64 || !c53.w_kcp || (W3 & 3)) return -1;
    const size_t n4s = std::max((size_t)PYR53_CG * H4 * W4, (size_t)4 * PYR53_CG * H5 * W5);      // the x4 planes' region holds the 1x1's four partial sums first
    const size_t lds = (((n4s + 3) & ~(size_t)3) + (((size_t)PYR53_CG * H5 * W5 + 3) & ~(size_t)3)) * sizeof(float) + 2 * (size_t)(W3 + H3) * sizeof(PyrCoef) + 2 * (size_t)W3 * sizeof(float2);
    if (lds > 64 * 1024 || (size_t)PYR53_CG * H3 * W3 >= (1u << 20)) return -1;      // (the kernel's index arithmetic: float4 indices below 2^18)
    static AttrMask attr = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(pyramid53_kernel<PYR53_CG>), 64 * 1024, attr);
    pyramid53_kernel<PYR53_CG><<<B * (64 / PYR53_CG), 256, lds, st>>>(x3, x4, y5, c53.w_kcp, c53.bias, c53.relu, out, H3, W3, H4, W4, H5, W5, 0);
    return 0;
}

void launch_pyramid_sum(const float* x3, const float* x4, const float* x5, float* out, int planes,
                        int H3, int W3, int H4, int W4, int H5, int W5, hipStream_t st) {
    const size_t lds = ((((size_t)H4 * W4 + (size_t)H5 * W5 + 3) & ~(size_t)3) * sizeof(float)) + 2 * (size_t)(W3 + H3) * sizeof(PyrCoef);      // planes + coefficient tables
    if (lds <= 64 * 1024)
        pyramid_sum_kernel<true><<<planes, 256, lds, st>>>(x3, x4, x5, out, H3, W3, H4, W4, H5, W5);
    else
        pyramid_sum_kernel<false><<<planes, 256, 0, st>>>(x3, x4, x5, out, H3, W3, H4, W4, H5, W5);
}

}  // namespace xfh
