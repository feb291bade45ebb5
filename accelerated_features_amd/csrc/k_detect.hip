// Sparse detection: 5x5 NMS + ordered compaction, scoring, top-k, descriptor sampling.
//   XFeat.NMS                         modules/xfeat.py:249-263
//   scores / argsort / top-k          modules/xfeat.py:77-87
//   InterpolateSparse2d (3 modes)     modules/interpolator.py:17-33  (arithmetic: SURVEY App. A.6)
//   descriptors                       modules/xfeat.py:70,90-93
//   extractDense top-k + gather       modules/xfeat.py:362-375
//
// Integer results are decided here, so the fp32 arithmetic that feeds a comparison follows the
// reference's operation order exactly (coordinate normalisation, fused un-normalise, round-
// half-even for 'nearest').  Ragged per-image lists are kept at fixed capacity with device
// counts; ordering uses wave ballots + block scans (row-major order is preserved).
#include "../../include/xfeat_hip.h"
#include "kernels.hpp"
#include <type_traits>

namespace xfh {

// ------------------------------------------------------------------------------------------
// NMS flags: pixel is kept iff heat > thr and no pixel of its 5x5 window (implicit -inf
// padding) is larger (== local max; plateaus keep every equal pixel).
// 1-D grid, XCD-grouped by image.
// ------------------------------------------------------------------------------------------
// Round 3: no LDS tile, no barrier.  A WAVE owns 64 columns x NMS_RB rows: every lane loads its column of the NMS_RB + 4 input rows (all loads in
// flight at once), lanes 0..3 the four halo columns; the 5 x 5 maximum is a vertical v_max3 pair per row and a horizontal pass through
// whole-wave DPP shifts (wave_shr / wave_shl by one lane, twice), the halo maxima entering at lanes 0 / 63 as the shifts' `old` operand.
// ~22 vector instructions per 64 pixels; the LDS version (36 x 68 tile, five LDS reads per horizontal maximum, a barrier per tile) ran at
// 2.5 TB/s.  Any width (scalar loads).
constexpr int NMS_RB = 16, NMS_TH = NMS_RB;
__device__ inline float dpp_wave_shr1(float v, float old) {      // lane i <- lane i - 1, lane 0 <- old
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ inline float dpp_wave_shl1(float v, float old) {      // lane i <- lane i + 1, lane 63 <- old
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__global__ __launch_bounds__(256) void nms_flags_kernel(const float* __restrict__ heat, int B, int H, int W, int WPR, int HT, float thr,
                                                        unsigned long long* __restrict__ mask, int* __restrict__ wcount) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // all strips of an image on one XCD: the 2-pixel halos of neighbouring strips hit that XCD's L2
    int b, item;
    if (!xcd_group_map(blockIdx.x, ceil_div(WPR * HT, 4), B, b, item)) return;
    const int unit = item * 4 + wave;
    if (unit >= WPR * HT) return;
    const int word = unit % WPR, y0 = (unit / WPR) * NMS_RB;
    const int x = word * 64 + lane;
    // halo columns: lane 0 -> x0 - 2, 1 -> x0 - 1, 2 -> x0 + 64, 3 -> x0 + 65 (other lanes: out of range = -inf)
    const int hx = lane < 2 ? word * 64 - 2 + lane : (lane < 4 ? word * 64 + 62 + lane : -1);
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(heat + (size_t)b * H * W), 0, H * W * 4, 0x00020000);
    const bool xin = x < W, hin = hx >= 0 && hx < W;
    float v[NMS_RB + 4], h[NMS_RB + 4];
#pragma unroll
    for (int r = 0; r < NMS_RB + 4; ++r) {
        const int gy = y0 - 2 + r;
        const bool yin = gy >= 0 && gy < H;
        // (out-of-image: an out-of-range offset would read 0, the padding value is -inf: select after the load)
        const float a = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rh, (yin && xin) ? (gy * W + x) * 4 : 0, 0, 0));
        const float c = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rh, (yin && hin) ? (gy * W + hx) * 4 : 0, 0, 0));
        v[r] = (yin && xin) ? a : -INFINITY;
        h[r] = (yin && hin) ? c : -INFINITY;
    }
#pragma unroll
    for (int r = 0; r < NMS_RB; ++r) {
        const int y = y0 + r;
        const float vm = fmaxf(fmaxf(fmaxf(v[r], v[r + 1]), v[r + 2]), fmaxf(v[r + 3], v[r + 4]));
        const float hm = fmaxf(fmaxf(fmaxf(h[r], h[r + 1]), h[r + 2]), fmaxf(h[r + 3], h[r + 4]));      // lanes 0..3: the halo columns' vertical maxima
        const float h1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hm), 1)), h2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hm), 2)),
                    h3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hm), 3));
        const float l1 = dpp_wave_shr1(vm, h1);            // column x - 1 (lane 0: x0 - 1)
        const float l2 = dpp_wave_shr1(l1, hm);            // column x - 2 (lane 0: x0 - 2 = its own halo value; lane 1: x0 - 1 via l1's lane 0)
        const float r1 = dpp_wave_shl1(vm, h2);            // column x + 1 (lane 63: x0 + 64)
        const float r2 = dpp_wave_shl1(r1, h3);            // column x + 2 (lane 63: x0 + 65; lane 62: x0 + 64 via r1's lane 63)
        const float m = fmaxf(fmaxf(fmaxf(vm, l1), l2), fmaxf(r1, r2));
        const float c = v[r + 2];
        const bool cand = (y < H) && xin && (c > thr) && (c == m);
        const unsigned long long bal = __ballot(cand);
        if (lane == 0 && y < H) {
            const size_t o = ((size_t)b * H + y) * WPR + word;
            mask[o] = bal;
            wcount[o] = __popcll(bal);
        }
    }
}

// Round 6: the same kernel for thr >= 0 (the hot path: detection_threshold 0.05), in 40 % of the vector instructions -- the PMC said the kernel above is bound by them (VALU
// active 93 % of its cycles, 51 instructions per 64-pixel row: canonicalising v_max pairs instead of v_max3, 260 v_readlane / v_writelane per wave of SPILLED scalars -- the
// per-row image-border masks --, a value select behind every load, a DPP move + a copy per shift).  What changes:
//   * a pixel outside the image may read as 0 instead of -inf: a candidate is > thr >= 0, so a zero in its window never equals it and never exceeds it -- the flags are
//     the same for ANY input values.  Rows outside the image are then simply out of the buffer's range (the hardware returns 0), columns outside carry an out-of-range
//     offset: no mask, no select, one address add per load;
//   * the halo columns sit in lanes 0, 1 (x0 - 2, x0 - 1) and 62, 63 (x0 + 64, x0 + 65); their contribution to lanes 0 / 1 / 62 / 63 is two DPP-shifted maxima
//     and a select, without read-lanes;
//   * the horizontal maximum is four v_max_f32 with a wave_shr:1 / wave_shl:1 operand (lanes beyond the wave read 0) on top of the vertical v_max3 pair;
//   * lane r collects row r's ballot: two stores per wave instead of two per row.
__device__ inline float vmax3(float a, float b, float c) {      // (no canonicalisation: the operands are plain loads; a NaN pixel is unspecified anyway, xfeat_hip.h)
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// the horizontal part of one row in ONE statement (the hazard rules of DPP operands -- a register a vector instruction wrote is readable by a DPP operand two
// instructions later -- are kept by the order inside it; the compiler does not look into it): in: vm, hm; out: u = max over columns x - 2 .. x, w = x .. x + 2 (lanes
// beyond the wave read 0), hm = the halo columns' part for lanes 0, 1 (row_mask 1: lanes 0-15) and 62, 63 (row_mask 8: lanes 48-63)
__device__ inline void nms_row_dpp(float vm, float& hm, float& u, float& w) {
    float t, s_;
    asm("s_nop 1\n\t"
        "v_max_f32_dpp %0, %5, %5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"      /* t  = columns x - 1, x */
        "v_max_f32_dpp %1, %5, %5 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"      /* s  = columns x, x + 1 */
        "v_max_f32_dpp %4, %4, %4 wave_shl:1 row_mask:0x1 bank_mask:0xf bound_ctrl:0\n\t"      /* hm: lane 0 <- max(h0, h1), lane 1 <- h1 (h2 = 0); lanes 16-63 keep their value */
        "v_max_f32_dpp %2, %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"      /* u  = columns x - 2 .. x   (t: two instructions old) */
        "v_max_f32_dpp %3, %1, %1 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"      /* w  = columns x .. x + 2 */
        "v_max_f32_dpp %4, %4, %4 wave_shr:1 row_mask:0x8 bank_mask:0xf bound_ctrl:0"            /* hm: lane 63 <- max(h63, h62), lane 62 <- h62 (h61 = 0)   (hm: two instructions old) */
        : "=&v"(t), "=&v"(s_), "=&v"(u), "=&v"(w), "+v"(hm) : "v"(vm));
}
template <int R>
__device__ inline void put_lane(unsigned& m, unsigned sv) {      // lane R of m <- the scalar sv
    asm("v_writelane_b32 %0, %1, %2" : "+v"(m) : "s"(sv), "n"(R));
}
__global__ __launch_bounds__(256) void nms_flags_zp_kernel(const float* __restrict__ heat, int B, int H, int W, int WPR, int HT, float thr,
                                                           unsigned long long* __restrict__ mask, int* __restrict__ wcount) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int b, item;
    if (!xcd_group_map(blockIdx.x, ceil_div(WPR * HT, 4), B, b, item)) return;
    const int unit = item * 4 + wave;
    if (unit >= WPR * HT) return;
    const int word = unit % WPR, y0 = (unit / WPR) * NMS_RB;
    const int x = word * 64 + lane;
    const int hx = lane < 2 ? word * 64 - 2 + lane : (lane >= 62 ? word * 64 + 2 + lane : -1);      // lanes 0, 1: x0 - 2, x0 - 1; lanes 62, 63: x0 + 64, x0 + 65
    const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(heat + (size_t)b * H * W), 0, H * W * 4, 0x00020000);
    // a lane's offset in row 0, or "far out of range" (stays out of range under every row offset added below: |row offset| < 2^30, launch check)
    const int vx = x < W ? x * 4 : (int)0x80000000, vh = (hx >= 0 && hx < W) ? hx * 4 : (int)0x80000000;
    float v[NMS_RB + 4], h[NMS_RB + 4];
#pragma unroll
    for (int r = 0; r < NMS_RB + 4; ++r) {
        const int ro = (y0 - 2 + r) * W * 4;             // (scalar; rows above / below the image: negative, or >= the buffer's size -> the load returns 0)
        v[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rh, vx + ro, 0, 0));
        h[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rh, vh + ro, 0, 0));
    }
    unsigned mlo = 0, mhi = 0;
    auto row = [&](auto RC) __attribute__((always_inline)) {
        constexpr int r = decltype(RC)::value;
        const float vm = vmax3(vmax3(v[r], v[r + 1], v[r + 2]), v[r + 3], v[r + 4]);
        float hm = vmax3(vmax3(h[r], h[r + 1], h[r + 2]), h[r + 3], h[r + 4]);      // lanes 0, 1, 62, 63: the halo columns' vertical maxima (0 elsewhere)
        float u, w;
        nms_row_dpp(vm, hm, u, w);
        const float m = vmax3(u, w, hm);
        const float c = v[r + 2];
        const unsigned long long bal = __ballot((c > thr) & (c == m));      // (columns >= W read 0: never > thr)
        // lane r keeps row r's ballot (v_writelane_b32: the ballot is already in scalar registers; the lane number is an immediate -- one scalar operand per instruction)
        if (lane == r) { mlo = (unsigned)bal; mhi = (unsigned)(bal >> 32); }
    };
    static_assert(NMS_RB == 16, "sixteen rows per wave, spelled out");
    row(std::integral_constant<int, 0>{}); row(std::integral_constant<int, 1>{}); row(std::integral_constant<int, 2>{}); row(std::integral_constant<int, 3>{});
    row(std::integral_constant<int, 4>{}); row(std::integral_constant<int, 5>{}); row(std::integral_constant<int, 6>{}); row(std::integral_constant<int, 7>{});
    row(std::integral_constant<int, 8>{}); row(std::integral_constant<int, 9>{}); row(std::integral_constant<int, 10>{}); row(std::integral_constant<int, 11>{});
    row(std::integral_constant<int, 12>{}); row(std::integral_constant<int, 13>{}); row(std::integral_constant<int, 14>{}); row(std::integral_constant<int, 15>{});
    const int y = y0 + lane;
    if (lane < NMS_RB && y < H) {
        const size_t o = ((size_t)b * H + y) * WPR + word;
        mask[o] = ((unsigned long long)mhi << 32) | mlo;
        wcount[o] = __popc(mlo) + __popc(mhi);
    }
}

// Any odd window (XFeat.NMS(kernel_size=k), xfeat.py:249-252: max_pool2d(k, stride 1, padding k/2)): one wave per 64-pixel mask
// word, every lane scans its own k x k window in global memory (helper API, not on the hot path which always uses 5).
__global__ __launch_bounds__(256) void nms_flags_generic_kernel(const float* __restrict__ heat, int B, int H, int W, int WPR, int rad, float thr,
                                                                unsigned long long* __restrict__ mask, int* __restrict__ wcount) {
    const int lane = threadIdx.x & 63;
    const size_t wid = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (size_t)B * H * WPR) return;
    const int word = (int)(wid % WPR);
    const int y = (int)((wid / WPR) % H), b = (int)(wid / ((size_t)WPR * H));
    const int x = word * 64 + lane;
    const float* hp = heat + (size_t)b * H * W;
    bool cand = false;
    if (x < W) {
        const float v = hp[(size_t)y * W + x];
        float m = v;
        for (int dy = -rad; dy <= rad; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -rad; dx <= rad; ++dx) {
                const int xx = x + dx;
                if (xx >= 0 && xx < W) m = fmaxf(m, hp[(size_t)yy * W + xx]);
            }
        }
        cand = (v > thr) && (v == m);
    }
    const unsigned long long bal = __ballot(cand);
    if (lane == 0) {
        mask[wid] = bal;
        wcount[wid] = __popcll(bal);
    }
}

// block-wide exclusive scan of one int per thread (1024 threads); returns the exclusive
// prefix, *total = block sum.  lds: >= 16 ints.
__device__ inline int block_exscan_1024(int v, int* lds, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int s = lds[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// grid (B), block 1024: scan the word counts, expand bits in row-major order
__global__ __launch_bounds__(1024) void nms_compact_kernel(const unsigned long long* __restrict__ mask,
                                                           const int* __restrict__ wcount, int H, int WPR, int cap,
                                                           unsigned* __restrict__ cand, int32_t* __restrict__ n_cand) {
    __shared__ int lds[16];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int NW = H * WPR;
    const int per = ceil_div(NW, 1024);
    const int beg = min(tid * per, NW), end = min(beg + per, NW);
    const int* wc = wcount + (size_t)b * NW;
    int local = 0;
    for (int i = beg; i < end; ++i) local += wc[i];
    int total;
    int off = block_exscan_1024(local, lds, &total);
    const unsigned long long* mk = mask + (size_t)b * NW;
    unsigned* out = cand + (size_t)b * cap;
    for (int i = beg; i < end; ++i) {
        unsigned long long m = mk[i];
        const int y = i / WPR, xb = (i - y * WPR) * 64;
        while (m) {
            const int bit = __ffsll((long long)m) - 1;
            m &= m - 1;
            if (off < cap) out[off] = ((unsigned)y << 16) | (unsigned)(xb + bit);
            ++off;
        }
    }
    if (tid == 0) n_cand[b] = total;
}

// ------------------------------------------------------------------------------------------
// sampling coordinate (interpolator.py:17-19 + ATen grid sampler, align_corners=False):
//   g1 = (2*(p/(S-1)) - 1) + 1   [fp32, reference order]   u = fma(g1, Sm/2, -0.5)
// ------------------------------------------------------------------------------------------
__device__ inline float sample_coord(int p, int S, int Sm) {
    const float q = (float)p / (float)(S - 1);
    const float g = 2.0f * q - 1.0f;      // 2*q is exact, so contraction cannot change this
    const float g1 = g + 1.0f;
    return __fmaf_rn(g1, (float)Sm * 0.5f, -0.5f);
}

__device__ inline float fetch0(const float* __restrict__ m, int Hm, int Wm, int y, int x) {
    return (x >= 0 && x < Wm && y >= 0 && y < Hm) ? m[(size_t)y * Wm + x] : 0.f;
}

// score = nearest(heat) * bilinear(reliability); (0,0) -> -1          (xfeat.py:77-80)
// key = (~ord(score) << 32) | slot : ascending key == descending score, ties by slot
struct ScoreSrc {
    const float* heat;      // (B,H,W)
    const float* rel;       // (B,H/8,W/8)
    const unsigned* cand;   // (B,cap)
    int H, W;
};
__device__ inline unsigned long long score_key(const ScoreSrc& s, int b, int cap, int i) {
    const unsigned c = s.cand[(size_t)b * cap + i];
    const int x = c & 0xffff, y = c >> 16;
    const int H = s.H, W = s.W, hc = H >> 3, wc = W >> 3;
    // nearest on the H x W heat map
    const float nx = sample_coord(x, W, W), ny = sample_coord(y, H, H);
    const int ix = (int)rintf(nx), iy = (int)rintf(ny);
    const float sn = fetch0(s.heat + (size_t)b * H * W, H, W, iy, ix);
    // bilinear on the (H/8) x (W/8) reliability map
    const float ux = sample_coord(x, W, wc), uy = sample_coord(y, H, hc);
    const float fx = floorf(ux), fy = floorf(uy);
    const float tx = ux - fx, ty = uy - fy;
    const int x0 = (int)fx, y0 = (int)fy;
    const float* rp = s.rel + (size_t)b * hc * wc;
    const float sb = fetch0(rp, hc, wc, y0, x0) * ((1.f - tx) * (1.f - ty)) + fetch0(rp, hc, wc, y0, x0 + 1) * (tx * (1.f - ty)) +
                     fetch0(rp, hc, wc, y0 + 1, x0) * ((1.f - tx) * ty) + fetch0(rp, hc, wc, y0 + 1, x0 + 1) * (tx * ty);
    float score = sn * sb;
    if (x == 0 && y == 0) score = -1.f;
    return ((unsigned long long)(~float_ord(score)) << 32) | (unsigned)i;
}

// ------------------------------------------------------------------------------------------
// per-image top-k of unique 64-bit keys (ascending) = argsort(-scores)[:top_k] (xfeat.py:83-87) / topk (xfeat.py:371),
// spread over the whole chip:
//   1. topk_sort_runs_kernel: workgroup (image, q) builds the keys of candidates [2048 q, 2048 q + 2048) (the score
//      sampling of xfeat.py:77-80 is fused here) and sorts them in LDS -> the key array of an image is a sequence of
//      sorted runs.  ~4 workgroups per VGA image instead of one.
//   2. topk_rank_merge_kernel: the final position of a key is its rank = its position in its own run + the number of
//      smaller keys in every other run (binary searches; the first 4 runs of the image -- n <= 8192 -- sit in LDS, later
//      runs are searched in global memory).  A key with rank < k is written straight to slot `rank` of the outputs: no
//      radix select, no global merge passes, any n and any top_k.  Both kernels launch 5 workgroups per image that stride
//      over the runs (a grid sized for the worst case n = H*W/8 costs more in empty 1024-thread workgroups than the work).
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// Ascending bitonic sort of n (power of two) 64-bit keys in LDS by a 1024-thread workgroup.  Every wave owns aligned
// blocks of 256 keys (4 consecutive keys per lane): all compare-exchange steps with stride < 256 stay inside the wave
// (registers for stride 1-2, v_permlane / ds_bpermute exchanges for 4..128) and need no workgroup barrier; only the
// strides >= 256 of the last log2(n/256) merge phases go through LDS with a barrier each (18 barriers at n = 4096
// instead of 78).
// ------------------------------------------------------------------------------------------
__device__ inline void cswap(unsigned long long& a, unsigned long long& b, bool up) {
    if ((a > b) == up) { const unsigned long long t = a; a = b; b = t; }
}
// Value of lane (lane ^ LS) without the LDS crossbar round trip of ds_bpermute (__shfl_xor): DPP quad permutes / row shifts / row rotate for
// 1, 2, 4, 8, the gfx950 permlane swaps for 16 and 32 -- plain VALU moves, no address register, no lgkmcnt wait.
template <int LS>
__device__ inline unsigned xchg_u32(unsigned v, int lane) {
    if constexpr (LS == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);          // quad_perm [1,0,3,2]
    else if constexpr (LS == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]
    else if constexpr (LS == 4) {
        const unsigned up = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0xF, true);               // row_shl:4: lane i <- i + 4
        const unsigned dn = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);               // row_shr:4: lane i <- i - 4
        return (lane & 4) ? dn : up;
    } else if constexpr (LS == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true);  // row_ror:8
    else if constexpr (LS == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);      // r[0] = rows {0,0,2,2}, r[1] = rows {1,1,3,3}
        return (lane & 16) ? r[0] : r[1];
    } else return xhalf_u32(v);
}
template <int LS>
__device__ inline unsigned long long xchg_u64(unsigned long long v, int lane) {
    return ((unsigned long long)xchg_u32<LS>((unsigned)(v >> 32), lane) << 32) | xchg_u32<LS>((unsigned)v, lane);
}
template <int LS>
__device__ inline void bitonic_lane_step(unsigned long long (&v)[4], int base, int lane, int size) {
    const bool lower = (lane & LS) == 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = base + 4 * lane + e;
        const bool up = (i & size) == 0;
        const unsigned long long other = xchg_u64<LS>(v[e], lane);
        // the lower index of the pair keeps the smaller key when the run ascends
        const bool keep_min = lower == up;
        v[e] = keep_min ? (v[e] < other ? v[e] : other) : (v[e] > other ? v[e] : other);
    }
}
// strides min(size/2, 128) .. 1 of the merge phase `size` on the wave's 4 keys per lane (block base index `base`)
__device__ inline void bitonic_wave_steps(unsigned long long (&v)[4], int base, int lane, int size) {
    const int top = min(size >> 1, 128);                           // partner lane distance = stride / 4
    if (top >= 128) bitonic_lane_step<32>(v, base, lane, size);
    if (top >= 64) bitonic_lane_step<16>(v, base, lane, size);
    if (top >= 32) bitonic_lane_step<8>(v, base, lane, size);
    if (top >= 16) bitonic_lane_step<4>(v, base, lane, size);
    if (top >= 8) bitonic_lane_step<2>(v, base, lane, size);
    if (top >= 4) bitonic_lane_step<1>(v, base, lane, size);
    const bool up = ((base + 4 * lane) & size) == 0;               // the 4 keys of a lane share the direction for size >= 4
    if (size >= 4) { cswap(v[0], v[2], up); cswap(v[1], v[3], up); }
    if (size >= 4) { cswap(v[0], v[1], up); cswap(v[2], v[3], up); }
    else { cswap(v[0], v[1], true); cswap(v[2], v[3], false); }    // size == 2: pairs alternate direction inside the lane
}
__device__ __forceinline__ void bitonic_sort_lds(unsigned long long* lk, int n) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (n < 256) {                                                  // tiny sorts: plain network (top_k < 256)
        for (int size = 2; size <= n; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int t = tid; t < (n >> 1); t += 1024) {
                    const int i = 2 * t - (t & (stride - 1)), j = i + stride;
                    const unsigned long long a = lk[i], c = lk[j];
                    if ((a > c) == ((i & size) == 0)) { lk[i] = c; lk[j] = a; }
                }
                __syncthreads();
            }
        return;
    }
    const int nblk = n >> 8;
    // phase 1: every 256-block fully sorted inside one wave (merge phases 2..256), direction by block parity
    for (int blk = wave; blk < nblk; blk += 16) {
        const int base = blk << 8;
        unsigned long long v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = lk[base + 4 * lane + e];
        for (int size = 2; size <= 256; size <<= 1) bitonic_wave_steps(v, base, lane, size);
#pragma unroll
        for (int e = 0; e < 4; ++e) lk[base + 4 * lane + e] = v[e];
    }
    __syncthreads();
    // phase 2: merge phases 512 .. n: strides >= 256 through LDS, the rest inside the waves
    for (int size = 512; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride >= 256; stride >>= 1) {
            for (int t = tid; t < (n >> 1); t += 1024) {
                const int i = 2 * t - (t & (stride - 1)), j = i + stride;
                const unsigned long long a = lk[i], c = lk[j];
                if ((a > c) == ((i & size) == 0)) { lk[i] = c; lk[j] = a; }
            }
            __syncthreads();
        }
        for (int blk = wave; blk < nblk; blk += 16) {
            const int base = blk << 8;
            unsigned long long v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = lk[base + 4 * lane + e];
            bitonic_wave_steps(v, base, lane, size);
#pragma unroll
            for (int e = 0; e < 4; ++e) lk[base + 4 * lane + e] = v[e];
        }
        __syncthreads();
    }
}

constexpr int TK_CH = 2048;          // keys per sorted run
constexpr int TK_LDS_RUNS = 4;       // runs of one image kept in LDS by the rank kernel (64 KB: two workgroups per CU); later runs are searched in global memory
constexpr int TK_WG_PER_IMAGE = 5;   // workgroups launched per image; each walks runs q, q + 5, ... (n <= 10240 is one run each)

// MODE 0: keys = score keys of the NMS candidates (ScoreSrc); MODE 1: keys = (~ord(vals[i]) << 32 | i)
template <int MODE>
__global__ __launch_bounds__(1024) void topk_sort_runs_kernel(ScoreSrc src, const float* __restrict__ vals, const int32_t* __restrict__ n_dev,
                                                              int n_const, int n_cap, int top_k, int qmax, int B,
                                                              unsigned long long* __restrict__ runs, int* __restrict__ nsel,
                                                              int32_t* __restrict__ n_valid) {
    __shared__ __attribute__((aligned(16))) unsigned long long lk[TK_CH];
    int b, q0;
    if (!xcd_group_map(blockIdx.x, qmax, B, b, q0)) return;
    const int tid = threadIdx.x;
    const int n = n_dev ? min(n_dev[b], n_cap) : n_const;
    if (q0 == 0 && tid == 0) {
        nsel[b] = min(top_k, n);
        if (n_valid) n_valid[b] = 0;
    }
    for (int q = q0; q * TK_CH < n; q += qmax) {
        const int base = q * TK_CH;
        const int csz = min(TK_CH, n - base);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int j = tid + e * 1024, i = base + j;
            unsigned long long key = ~0ull;
            if (j < csz) {
                if constexpr (MODE == 0) key = score_key(src, b, n_cap, i);
                else key = ((unsigned long long)(~float_ord(vals[(size_t)b * n_cap + i])) << 32) | (unsigned)i;
            }
            lk[j] = key;
        }
        __syncthreads();
        // short runs sort a smaller power of two (the padding keys ~0 sort last and are not written back)
        int npad = 256;
        while (npad < csz) npad <<= 1;
        bitonic_sort_lds(lk, npad);
        unsigned long long* out = runs + (size_t)b * n_cap + base;
        for (int j = tid; j < csz; j += 1024) out[j] = lk[j];
        __syncthreads();
    }
}

// number of keys < x in the ascending run a[0..len), len <= TK_CH
__device__ inline int run_lower_bound(const unsigned long long* a, int len, unsigned long long x) {
    int pos = 0;
#pragma unroll
    for (int step = TK_CH; step > 0; step >>= 1) {
        const int t = pos + step;
        if (t <= len && a[t - 1] < x) pos = t;
    }
    return pos;
}

// Workgroup (image, q0) ranks the keys of runs q0, q0 + nq, ... and writes each key with rank < k to slot `rank` of the sorted
// key list `skeys` (ONE 8-byte scattered store per key; a first version scattered sel / kpts / scores rows from here and spent
// 13 of its 25 us on those partial-line writes -- the consumers, which walk the list in rank order, now derive them).
__global__ __launch_bounds__(1024) void topk_rank_merge_kernel(const unsigned long long* __restrict__ runs, const int32_t* __restrict__ n_dev,
                                                               int n_const, int n_cap, int top_k, int nq, int B, int lds_runs,
                                                               unsigned long long* __restrict__ skeys) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long lr[];
    int b, q0;
    if (!xcd_group_map(blockIdx.x, nq, B, b, q0)) return;
    const int tid = threadIdx.x;
    const int n = n_dev ? min(n_dev[b], n_cap) : n_const;
    const int k = min(top_k, n);
    if (q0 * TK_CH >= n) return;
    const int nruns = ceil_div(n, TK_CH);
    const unsigned long long* gr = runs + (size_t)b * n_cap;
    const int nl = min(n, lds_runs * TK_CH);              // keys [0, nl) = the first lds_runs runs live in LDS
    {
        unsigned long long t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = tid + u * 1024; t[u] = i < nl ? gr[i] : 0ull; }       // independent loads in flight
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = tid + u * 1024; if (i < nl) lr[i] = t[u]; }
        for (int i = tid + 8192; i < nl; i += 1024) lr[i] = gr[i];
    }
    __syncthreads();
    for (int q = q0; q * TK_CH < n; q += nq) {
        const int base = q * TK_CH;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int j = tid + e * 1024, i = base + j;
            if (i < n && j < k) {                          // rank >= position in the own run
                const unsigned long long key = i < nl ? lr[i] : gr[i];
                int rank = j;
                for (int r = 0; r < nruns && rank < k; ++r) {
                    if (r == q) continue;
                    const int len = min(TK_CH, n - r * TK_CH);
                    rank += r < lds_runs ? run_lower_bound(lr + r * TK_CH, len, key) : run_lower_bound(gr + r * TK_CH, len, key);
                }
                if (rank < k) skeys[(size_t)b * top_k + rank] = key;
            }
        }
    }
}

// keys/runs: (B, n_cap) u64 scratch.  Either `src` (sparse path) or `vals` (B, n_cap floats) feeds the keys.
// Result: skeys (B, top_k) ascending keys (entries [nsel[b], top_k) are left untouched), nsel (B) = min(top_k, n).
static void run_topk(const ScoreSrc* src, const float* vals, unsigned long long* runs, const int32_t* n_dev, int n_const, int n_cap,
                     int top_k, int B, unsigned long long* skeys, int* nsel, int32_t* n_valid, hipStream_t st) {
    const int qmax = min(ceil_div(n_cap, TK_CH), TK_WG_PER_IMAGE);          // workgroups per image (each strides over the runs)
    if (src)
        topk_sort_runs_kernel<0><<<xcd_grid_size(qmax, B), 1024, 0, st>>>(*src, nullptr, n_dev, n_const, n_cap, top_k, qmax, B, runs, nsel, n_valid);
    else
        topk_sort_runs_kernel<1><<<xcd_grid_size(qmax, B), 1024, 0, st>>>(ScoreSrc{}, vals, n_dev, n_const, n_cap, top_k, qmax, B, runs, nsel, n_valid);
    const int lds_runs = min(ceil_div(n_cap, TK_CH), TK_LDS_RUNS);
    static AttrMask attr_mask = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(topk_rank_merge_kernel), TK_LDS_RUNS * TK_CH * 8, attr_mask);
    topk_rank_merge_kernel<<<xcd_grid_size(qmax, B), 1024, (size_t)lds_runs * TK_CH * 8, st>>>(runs, n_dev, n_const, n_cap, top_k, qmax, B, lds_runs, skeys);
}

// ------------------------------------------------------------------------------------------
// 1 / max(||feats[pixel,:]||, 1e-12)   (F.normalize(M1, dim=1), xfeat.py:70), feats NHWC
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void invnorm_kernel(const float* __restrict__ feats, int npix, float* __restrict__ inv) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int pix = g >> 4, part = g & 15;
    float s = 0.f;
    if (pix < npix) {
        const float4 v = *reinterpret_cast<const float4*>(feats + (size_t)pix * 64 + part * 4);
        s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    s += __shfl_xor(s, 8, 64);
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 1, 64);
    if (pix < npix && part == 0) inv[pix] = 1.f / fmaxf(sqrtf(s), 1e-12f);
}

// Keys cubic-convolution weights, A = -0.75 (ATen get_cubic_upsample_coefficients)
__device__ inline void cubic_w(float t, float w[4]) {
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

// 16 lanes per selected key-point (4 channels each as one float4), 4 key-points per wave:
//   desc = normalize( bicubic( normalize(M1, dim=1) ) )                   (xfeat.py:70,90-93)
// The 16 taps are 16 independent 256-B row reads per key-point (1 KiB per wave-instruction).
// Also the epilogue of the top-k (xfeat.py:83-87,96-103): key-point j of the sorted key list -> kpts (x*rw, y*rh), score, and
// n_valid = number of returned points with score > 0 (a prefix, the list is sorted).
__global__ __launch_bounds__(256) void descriptor_kernel(const float* __restrict__ feats, const float* __restrict__ inv,
                                                         const unsigned* __restrict__ cand, const unsigned long long* __restrict__ skeys,
                                                         const int* __restrict__ nsel, int H, int W, int cap, int top_k,
                                                         int B, int blocks_per_img, float rw, float rh, float* __restrict__ kpts,
                                                         float* __restrict__ scores, int32_t* __restrict__ n_valid, float* __restrict__ desc,
                                                         uint16_t* __restrict__ desc16) {
    const int sub = threadIdx.x & 15;
    int b, blk;                       // all key-points of an image on one XCD: its feats stay in that L2
    if (!xcd_group_map(blockIdx.x, blocks_per_img, B, b, blk)) return;
    const int j = blk * 16 + (threadIdx.x >> 4);
    if (j >= top_k) return;
    float4* dp = reinterpret_cast<float4*>(desc + ((size_t)b * top_k + j) * 64) + sub;
    const int k = nsel[b];
    ushort4* dp16 = desc16 ? reinterpret_cast<ushort4*>(desc16 + ((size_t)b * top_k + j) * 64) + sub : nullptr;
    if (j >= k) {                     // fixed-capacity outputs: rows past the list are zero
        *dp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (dp16) *dp16 = make_ushort4(0, 0, 0, 0);
        if (sub == 0) {
            *reinterpret_cast<float2*>(kpts + ((size_t)b * top_k + j) * 2) = make_float2(0.f, 0.f);
            scores[(size_t)b * top_k + j] = 0.f;
        }
        return;
    }
    const unsigned long long key = skeys[(size_t)b * top_k + j];
    const unsigned c = cand[(size_t)b * cap + (unsigned)(key & 0xffffffffu)];
    const int x = c & 0xffff, y = c >> 16;
    if (sub == 0) {
        const float score = ord_float(~(unsigned)(key >> 32));
        *reinterpret_cast<float2*>(kpts + ((size_t)b * top_k + j) * 2) = make_float2((float)x * rw, (float)y * rh);
        scores[(size_t)b * top_k + j] = score;
        if (score > 0.f) {            // the last positive score ends the valid prefix (n_valid was zeroed by the sort kernel)
            const bool last = (j + 1 >= k) || !(ord_float(~(unsigned)(skeys[(size_t)b * top_k + j + 1] >> 32)) > 0.f);
            if (last) n_valid[b] = j + 1;
        }
    }
    const int hc = H >> 3, wc = W >> 3;
    const float ux = sample_coord(x, W, wc), uy = sample_coord(y, H, hc);
    const float fx = floorf(ux), fy = floorf(uy);
    const int x0 = (int)fx - 1, y0 = (int)fy - 1;
    // Lane `sub` of a key-point's 16 lanes prepares TAP sub = (row sub >> 2, column sub & 3): its cubic weights, its 1 / |feats| (one load instruction per key-point group
    // instead of sixteen broadcast loads), their product w = (wx wy) / |feats|, and the byte offset of its pixel.  A tap outside the map has w = 0 and a clamped (valid)
    // address: 0 x finite = 0, the value a skipped tap leaves.  Every lane then needs every tap's (offset, w): both arrive as DPP OPERANDS (row_newbcast: lane t of the 16
    // to all of them) of the instruction that uses them -- v_add_u32_dpp for the address, four v_fmac_f32_dpp per tap for its 16-byte slice of the row: 5 vector instructions
    // per tap.  The round-6 counters had this kernel at 60 % vector-ALU busy with 353 instructions per wave (per tap: a DPP move, an address shift-or + add, four multiplications
    // by 1 / |feats| and four fused multiply-adds, + four IEEE divisions by the norm, + four LDS round trips for its sum); it is bound by instruction count, not by misses.
    const int tr = sub >> 2, ti = sub & 3;
    float wxy;
    {
        // Keys cubic-convolution weight of tap offset i in 0 .. 3 at fraction t: the same operations as cubic_w on x = t + 1, t, 1 - t, 2 - t
        auto cubic1 = [](float t, int i) {
            const float A = -0.75f;
            const float x = i == 0 ? t + 1.f : i == 1 ? t : i == 2 ? 1.f - t : 2.f - t;
            const float outer = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
            const float inner = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
            return (i == 0 || i == 3) ? outer : inner;
        };
        wxy = cubic1(ux - fx, ti) * cubic1(uy - fy, tr);
    }
    const int npix = hc * wc;
    const __amdgpu_buffer_rsrc_t rf = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(feats + (size_t)b * npix * 64), 0, npix * 256, 0x00020000);
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(inv + (size_t)b * npix), 0, npix * 4, 0x00020000);
    const int ty = y0 + tr, tx = x0 + ti;
    const bool t_in = (unsigned)ty < (unsigned)hc && (unsigned)tx < (unsigned)wc;
    const int pix_mine = min(max(ty, 0), hc - 1) * wc + min(max(tx, 0), wc - 1);
    const float inv_mine = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ri, pix_mine * 4, 0, 0));
    const float w_mine = t_in ? wxy * inv_mine : 0.f;
    const int off_mine = pix_mine * 256;
    const int sub16 = sub * 16;
    typedef unsigned u32x4d __attribute__((ext_vector_type(4)));
    u32x4d tap[16];
#define XFH_TAP_LOAD(T) tap[T] = __builtin_bit_cast(u32x4d, __builtin_amdgcn_raw_buffer_load_b128(rf, (int)__builtin_amdgcn_update_dpp(0u, (unsigned)off_mine, 0x150 + (T), 0xf, 0xf, true) + sub16, 0, 0));
    XFH_TAP_LOAD(0) XFH_TAP_LOAD(1) XFH_TAP_LOAD(2) XFH_TAP_LOAD(3) XFH_TAP_LOAD(4) XFH_TAP_LOAD(5) XFH_TAP_LOAD(6) XFH_TAP_LOAD(7)
    XFH_TAP_LOAD(8) XFH_TAP_LOAD(9) XFH_TAP_LOAD(10) XFH_TAP_LOAD(11) XFH_TAP_LOAD(12) XFH_TAP_LOAD(13) XFH_TAP_LOAD(14) XFH_TAP_LOAD(15)
#undef XFH_TAP_LOAD
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#define XFH_TAP_FMA(T) { const float w_ = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(w_mine), 0x150 + (T), 0xf, 0xf, true)); \
        acc.x = fmaf(w_, __uint_as_float(tap[T].x), acc.x); acc.y = fmaf(w_, __uint_as_float(tap[T].y), acc.y); \
        acc.z = fmaf(w_, __uint_as_float(tap[T].z), acc.z); acc.w = fmaf(w_, __uint_as_float(tap[T].w), acc.w); }
    XFH_TAP_FMA(0) XFH_TAP_FMA(1) XFH_TAP_FMA(2) XFH_TAP_FMA(3) XFH_TAP_FMA(4) XFH_TAP_FMA(5) XFH_TAP_FMA(6) XFH_TAP_FMA(7)
    XFH_TAP_FMA(8) XFH_TAP_FMA(9) XFH_TAP_FMA(10) XFH_TAP_FMA(11) XFH_TAP_FMA(12) XFH_TAP_FMA(13) XFH_TAP_FMA(14) XFH_TAP_FMA(15)
#undef XFH_TAP_FMA
    // |desc|^2 over the key-point's 16 lanes: an xor butterfly on DPP operands (quad_perm 1032 / 2301, row_half_mirror, row_mirror: every lane ends with the same bits)
    float n2 = acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
#define XFH_DPP_ADD(v, ctrl) v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), ctrl, 0xf, 0xf, true))
    XFH_DPP_ADD(n2, 0xb1);       // quad_perm [1, 0, 3, 2]
    XFH_DPP_ADD(n2, 0x4e);       // quad_perm [2, 3, 0, 1]
    XFH_DPP_ADD(n2, 0x141);      // row_half_mirror
    XFH_DPP_ADD(n2, 0x140);      // row_mirror
#undef XFH_DPP_ADD
    const float rd = 1.f / fmaxf(sqrtf(n2), 1e-12f);      // (one IEEE division; the four quotients are products with it: within an ulp of acc / d)
    const float4 o = make_float4(acc.x * rd, acc.y * rd, acc.z * rd, acc.w * rd);
    *dp = o;
    if (dp16) {                       // fp16 copy of 256 * row (round-to-nearest-even) for the matcher's filter sweep: saves its conversion passes
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 q;
        q[0] = (_Float16)(o.x * 256.f); q[1] = (_Float16)(o.y * 256.f); q[2] = (_Float16)(o.z * 256.f); q[3] = (_Float16)(o.w * 256.f);
        *dp16 = __builtin_bit_cast(ushort4, q);
    }
}

void prof_begin(Profiler* p, int which, hipStream_t st);
void prof_end(Profiler* p, int which, hipStream_t st, double flops, double bytes);

// 5 x 5 NMS flags: thr >= 0 (every detection threshold) takes the zero-padding form, a negative threshold (XFeat.NMS on arbitrary data) the -inf-padding one
static void launch_nms_flags5(const float* heat, int B, int H, int W, int WPR, float thr, unsigned long long* mask, int* wcount, hipStream_t st) {
    const unsigned grid = xcd_grid_size(ceil_div(WPR * ceil_div(H, NMS_TH), 4), B);
    if (thr >= 0.f && (size_t)(H + 4) * W * 4 < (1u << 30))
        nms_flags_zp_kernel<<<grid, 256, 0, st>>>(heat, B, H, W, WPR, ceil_div(H, NMS_TH), thr, mask, wcount);
    else
        nms_flags_kernel<<<grid, 256, 0, st>>>(heat, B, H, W, WPR, ceil_div(H, NMS_TH), thr, mask, wcount);
}

void launch_detect(const DetectWs& ws, const float* heat, const float* reliab, const float* feats, const float* invnorm, int B, int H, int W,
                   float thr, int top_k, int cap, float rw, float rh, float* kpts, float* scores, float* desc,
                   int32_t* n_valid, int32_t* n_cand, hipStream_t st, uint16_t* desc16, Profiler* prof) {
    const int WPR = ceil_div(W, 64);
    const int hc = H / 8, wc = W / 8;
    prof_begin(prof, XFH_SPAN_NMS_FLAGS, st);
    launch_nms_flags5(heat, B, H, W, WPR, thr, ws.mask, ws.wcount, st);
    prof_end(prof, XFH_SPAN_NMS_FLAGS, st, 0, 0);
    prof_begin(prof, XFH_SPAN_NMS_COMPACT, st);
    nms_compact_kernel<<<B, 1024, 0, st>>>(ws.mask, ws.wcount, H, WPR, cap, ws.cand, n_cand);
    prof_end(prof, XFH_SPAN_NMS_COMPACT, st, 0, 0);
    const ScoreSrc src{heat, reliab, ws.cand, H, W};
    prof_begin(prof, XFH_SPAN_TOPK, st);
    run_topk(&src, nullptr, ws.keys, n_cand, 0, cap, top_k, B, ws.skeys, ws.nsel, n_valid, st);
    prof_end(prof, XFH_SPAN_TOPK, st, 0, 0);
    prof_begin(prof, XFH_SPAN_DESCRIPTOR, st);
    if (!invnorm) {           // not handed over by the backbone call: one pass over the feature maps
        invnorm_kernel<<<ceil_div(B * hc * wc * 16, 256), 256, 0, st>>>(feats, B * hc * wc, ws.invnorm);
        invnorm = ws.invnorm;
    }
    descriptor_kernel<<<xcd_grid_size(ceil_div(top_k, 16), B), 256, 0, st>>>(feats, invnorm, ws.cand, ws.skeys, ws.nsel, H, W,
                                                                            cap, top_k, B, ceil_div(top_k, 16), rw, rh, kpts, scores, n_valid, desc, desc16);
    prof_end(prof, XFH_SPAN_DESCRIPTOR, st, 0, 0);
}

// stand-alone NMS (XFeat.NMS): flags + compaction + int64 (x,y) list, zero padded
__global__ __launch_bounds__(256) void cand_to_xy_kernel(const unsigned* __restrict__ cand, const int32_t* __restrict__ n_cand,
                                                         int cap, int64_t* __restrict__ xy) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cap) return;
    int64_t x = 0, y = 0;
    if (i < min(n_cand[b], cap)) {
        const unsigned c = cand[(size_t)b * cap + i];
        x = c & 0xffff; y = c >> 16;
    }
    xy[((size_t)b * cap + i) * 2 + 0] = x;
    xy[((size_t)b * cap + i) * 2 + 1] = y;
}

void launch_nms_only(const DetectWs& ws, const float* heat, int B, int H, int W, float thr, int kernel_size, int cap, int64_t* xy,
                     int32_t* n_cand, hipStream_t st) {
    const int WPR = ceil_div(W, 64);
    // 5 x 5 (the hot path's window): the register / DPP kernel, any width; every other window goes to the per-pixel kernel
    if (kernel_size == 5)
        launch_nms_flags5(heat, B, H, W, WPR, thr, ws.mask, ws.wcount, st);
    else
        nms_flags_generic_kernel<<<(unsigned)(((size_t)B * H * WPR + 3) / 4), 256, 0, st>>>(heat, B, H, W, WPR, kernel_size / 2, thr, ws.mask, ws.wcount);
    nms_compact_kernel<<<B, 1024, 0, st>>>(ws.mask, ws.wcount, H, WPR, cap, ws.cand, n_cand);
    cand_to_xy_kernel<<<dim3(ceil_div(cap, 256), B), 256, 0, st>>>(ws.cand, n_cand, cap, xy);
}

// ------------------------------------------------------------------------------------------
// plain top-k (descending values, ties: lower index first) for extractDense (xfeat.py:371)
// ------------------------------------------------------------------------------------------
void launch_topk_desc(const float* vals, int B, int n, int k, unsigned long long* keys, unsigned long long* skeys, int* nsel,
                      hipStream_t st) {
    run_topk(nullptr, vals, keys, nullptr, n, n, k, B, skeys, nsel, nullptr, st);
}

// wave per selected cell: raw features + corner coordinates                (xfeat.py:366-375,388)
__global__ __launch_bounds__(256) void dense_gather_kernel(const float* __restrict__ feats, const unsigned long long* __restrict__ skeys,
                                                           int hc, int wc, int k, float rw, float rh, float scale_div,
                                                           float* __restrict__ kpts, float* __restrict__ desc,
                                                           int32_t* __restrict__ cell_index) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (j >= k) return;
    const unsigned cell = (unsigned)(skeys[(size_t)b * k + j] & 0xffffffffu);
    desc[((size_t)b * k + j) * 64 + lane] = feats[((size_t)b * hc * wc + cell) * 64 + lane];
    if (lane == 0) {
        const int ci = cell / wc, cj = cell - ci * wc;
        kpts[((size_t)b * k + j) * 2 + 0] = ((float)(cj * 8) * rw) / scale_div;
        kpts[((size_t)b * k + j) * 2 + 1] = ((float)(ci * 8) * rh) / scale_div;
        if (cell_index) cell_index[(size_t)b * k + j] = (int32_t)cell;
    }
}

void launch_dense_gather(const float* feats, const unsigned long long* skeys, int B, int hc, int wc, int k, float rw, float rh,
                         float scale_div, float* kpts, float* desc, int32_t* cell_index, hipStream_t st) {
    dense_gather_kernel<<<dim3(ceil_div(k, 4), B), 256, 0, st>>>(feats, skeys, hc, wc, k, rw, rh, scale_div, kpts, desc,
                                                                 cell_index);
}

}  // namespace xfh
