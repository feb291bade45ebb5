// block1 + skip1 fused (the body of block1_fused_kernel / block1_mx_kernel, k_conv_direct.hip).  A header of its own so that tests/emu/ can compile the SAME source
// for the host (XFH_HOST_EMU: 512 host threads per workgroup, LDS as a buffer, the matrix instruction and the LDS-DMA emulated) and run it against a convolution
// reference without a GPU -- index arithmetic, tile layouts and barriers of a kernel that is new are checked before it meets the hardware.
#pragma once
#ifndef XFH_HOST_EMU
#include "kernels.hpp"
#include "bx_split.hpp"
#include <type_traits>
#ifndef XFH_DYN_LDS
#define XFH_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) float name[]
#endif
#define XFH_NOP16_2(a, b) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b))              /* 16 idle slots behind an MFMA group, tied to its accumulators */
#define XFH_NOP16_3(a, b, c) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b), "+v"(c))
#ifndef XFH_PIN
#define XFH_PIN(x) asm volatile("" : "+v"(x))                                                 /* the value is computed HERE */
#endif
#ifndef XFH_GPTR_DEFINED
#define XFH_GPTR_DEFINED
typedef __attribute__((address_space(1))) const void* xfh_gptr_t;
typedef __attribute__((address_space(3))) void* xfh_lptr_t;
#endif
#else
#include "bx_split.hpp"
#endif
#include "block1_fx.hpp"

namespace xfh {

// ------------------------------------------------------------------------------------------
// block1 fused: gray (B,1,H,W) -> x1 = block1(gray) + skip1(gray)  (B,24,H/4,W/4)
//   (modules/model.py:40-48,140).  One workgroup = 8 x 16 output pixels.  The four
//   low-channel layers run back to back on LDS-resident tiles (halo recomputed per tile,
//   ~15 % extra FMAs), so the 4/8/8-channel full- and half-resolution activations
//   (19.7 MB/frame written and re-read by the layer-at-a-time version) never reach HBM:
//   the kernel reads the gray tile once and writes x1 once.
//
//   tile extents (rows x cols), origin in its own map:
//     out  8 x 16  at (Y4, X4)            [H/4 x W/4]
//     c3  17 x 33  at (2Y4-1, 2X4-1)      [H/2 x W/2]   conv3 8->8 s1
//     c2  19 x 35  at (2Y4-2, 2X4-2)      [H/2 x W/2]   conv2 4->8 s2
//     c1  39 x 71  at (4Y4-5, 4X4-5)      [H x W]       conv1 1->4 s1 (never stored: recomputed inside conv2)
//     g   41 x 73  at (4Y4-6, 4X4-6)      [H x W]       normalised gray
//   Positions outside a map are stored as 0 = the next conv's zero padding.
//   Weights are read with wave-uniform addresses (scalar loads, SGPR operands of v_fmac).
// ------------------------------------------------------------------------------------------
namespace b1 {
constexpr int OH = 8, OW = 16;
constexpr int C3H = 17, C3W = 33, C2H = 19, C2W = 35, C1H = 39, C1W = 71, GH = 41, GW = 73;
constexpr int G_OFF = 0, G_SZ = GH * GW;                  // 2993 (+ 1: rows are loaded as column pairs, the last pair of the last row spills one element)
constexpr int SK_OFF = G_OFF + G_SZ + 3, SK_SZ = OH * OW; // 4 x 4 averages of the gray tile (skip1's AvgPool2d), one per output pixel  (+ 3: the spill element, and every tile behind it 16-byte aligned)
constexpr int C2_SZ = 8 * C2H * C2W;                      // 5320
// mode 5 (vector ALUs only; conv1 recomputed inside conv2, no c1 tile): g | sk | c2 | c3 = 51.7 KB -> three workgroups per CU.  (c3 over the dead gray tile
// = 39.7 KB = four per CU measured slower: more waves than the LDS pipe and L1 feed.)
constexpr int F_SK_OFF = SK_OFF, F_C2_OFF = SK_OFF + SK_SZ, F_C3_OFF = F_C2_OFF + C2_SZ, F_LDS_FLOATS = F_C3_OFF + 8 * C3H * C3W;      // 12930 floats
// mode 7 (conv3, conv4 on the matrix cores): no skip table -- every thread of stage 4 holds its pixel's average in a register -- so the tiles move up by it and conv3's weight image
// fits behind them in 53 616 bytes: three workgroups per CU also if LDS is handed out in 1280-byte granules (42 of them; 160 KB / 3 = 54 613 bytes)
constexpr int M_C2_OFF = SK_OFF, M_C3_OFF = M_C2_OFF + C2_SZ, M_LDS_FLOATS = M_C3_OFF + 8 * C3H * C3W;      // 12804 floats
static_assert(SK_OFF % 4 == 0 && F_C2_OFF % 4 == 0 && F_C3_OFF % 4 == 0 && F_LDS_FLOATS % 4 == 0 && M_C3_OFF % 4 == 0 && M_LDS_FLOATS % 4 == 0,
              "16-byte aligned tiles: the fp16-pair planes of mode 7 are read as b128");
}  // namespace b1

template <int MODE>      // 5 = every layer on the vector ALUs (fp32's range: the fallback of the range guard); 7 = conv3 and conv4 on the fp16 matrix cores (the default).
                         // Both recompute conv1 from the gray tile inside conv2 (no c1 tile in LDS)
__device__ __forceinline__ void block1_fused_body(const float* __restrict__ gray, const float* __restrict__ coef, float* __restrict__ x1, int B, int H, int W,
                                                           int tiles_x, int tiles_y,
                                                           const float* __restrict__ w1, const float* __restrict__ bb1,
                                                           const float* __restrict__ w2, const float* __restrict__ bb2,
                                                           const float* __restrict__ w3, const float* __restrict__ bb3,
                                                           const float* __restrict__ w4, const float* __restrict__ bb4,
                                                           const float* __restrict__ skw, const float* __restrict__ skb,
                                                           const void* __restrict__ w4fx, const void* __restrict__ w3fx, int* __restrict__ status, int cold) {
    using namespace b1;
    static_assert(MODE == 5 || MODE == 7, "block1: 5 (vector ALUs) or 7 (conv3, conv4 on the matrix cores)");
    constexpr bool MX = MODE == 7;                       // conv4 (8 -> 24, stride 2) as 18 v_mfma_f32_16x16x32_f16 per wave in the fp16-pair arithmetic (block1_fx.hpp):
                                                         // stage 3 leaves c3 as fp16 pairs, the compact weight image lands over the dead gray tile during stage 3;
                                                         // conv3 (8 -> 8) too: stage 2 leaves c2 as fp16 pairs, 19 blocks of 16 pixel pairs x 9 MFMAs over the 8 waves; its weight
                                                         // image (2.3 KB behind the tiles) is fetched when the kernel starts
    if constexpr (MX) kernel_entry_hooks(cold);          // debug: code-position shift / cold instruction cache (common.hpp)
    XFH_DYN_LDS(lds);
    typedef xfh_gptr_t gptr_t;
    typedef xfh_lptr_t lptr_t;
    if constexpr (MX) {      // three 1-KiB pieces, the last one 288 bytes = 18 lanes
        const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), ln64 = threadIdx.x & 63;
        if (wv < 3 && (wv < 2 || ln64 < (b1fx::W3_BYTES - 2048) / 16))
            __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const unsigned char*>(w3fx) + wv * 1024 + ln64 * 16),
                                             (lptr_t)(reinterpret_cast<unsigned char*>(lds + M_LDS_FLOATS) + wv * 1024), 16, 0, 0);
    }
    float* G = lds + G_OFF;
    float* SK = lds + F_SK_OFF;
    float* C2 = lds + (MX ? M_C2_OFF : F_C2_OFF);
    float* C3 = lds + (MX ? M_C3_OFF : F_C3_OFF);
    const int tid = threadIdx.x;
    // the tiles of an image run on one XCD: the 4-pixel gray halos of neighbouring tiles hit its L2
    int b, item;
    if (!xcd_group_map(blockIdx.x, tiles_x * tiles_y, B, b, item)) return;
    const int Y4 = (item / tiles_x) * OH, X4 = (item % tiles_x) * OW;
    const int H2 = H >> 1, W2 = W >> 1, H4 = H >> 2, W4 = W >> 2;
    const float* gb = gray + (size_t)b * H * W;

    // ---- stage 0: gray tile, instance-normalised on the way in (zero padding stays zero) -------
    const float alpha = coef[2 * b], beta = coef[2 * b + 1];
    {   // the tile as 8-byte column pairs (its origin 4 X4 - 6 and W are even: a pair never straddles the image border), all three loads of a
        // thread in flight together (a rolled loop waits for each one); half the index arithmetic of the dword version (PMC: the kernel is bound
        // by the number of vector instructions it issues, and 46 % of them are not FMAs)
        constexpr int PW = (GW + 1) / 2, NP = GH * PW, NL = (NP + 511) / 512;      // 37 pairs per row (the last one holds column 72 and a spill)
        float2 raw[NL];
        bool in[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int e = tid + k * 512;
            const int r = e / PW, c = 2 * (e - r * PW);
            const int gy = 4 * Y4 - 6 + r, gx = 4 * X4 - 6 + c;
            in[k] = e < NP && gy >= 0 && gy < H && gx >= 0 && gx < W;
            raw[k] = in[k] ? *reinterpret_cast<const float2*>(gb + (size_t)gy * W + gx) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int e = tid + k * 512;
            if (e < NP) {
                const int r = e / PW, c = 2 * (e - r * PW);
                float* g = G + r * GW + c;                   // (row pitch 73: odd rows are only 4-byte aligned -> two dword stores)
                g[0] = in[k] ? fmaf(raw[k].x, alpha, beta) : 0.f;
                if (c + 1 < GW || r + 1 == GH) g[1] = in[k] ? fmaf(raw[k].y, alpha, beta) : 0.f;      // (column 73 of rows 0..39 is column 0 of the next row: its own pair writes it)
            }
        }
    }
    if constexpr (MX) lds_dma_barrier(); else __syncthreads();      // (MX: every barrier of a kernel with LDS-DMA waits for it -- tools/check_dma_barriers.py; nothing is in flight that is not needed here)

    // ---- (stage 1, conv1 1->4 s1, has no pass of its own: stage 2 recomputes the nine c1 pixels of its window from the gray tile) ----
    float sk_reg = 0.f;     // MX: skip1's 4 x 4 average of THIS thread's stage-4 pixel (wave = output row, lane & 15 = column): the four lanes that share a pixel sum a
                            // window row each and exchange (two wave shuffles); no table in LDS
    if constexpr (MX) {
        const int r = tid >> 6, c = tid & 15, i = (tid >> 4) & 3;
        const float* g4 = G + (4 * r + 6 + i) * GW + 4 * c + 6;
        float sm = (g4[0] + g4[1]) + (g4[2] + g4[3]);
        sm += __shfl_xor(sm, 16, 64);
        sm += __shfl_xor(sm, 32, 64);
        sk_reg = sm * 0.0625f;
    } else
    if (tid >= 384) {       // skip1's 4 x 4 averages, once per output pixel
        const int p = tid - 384, r = p >> 4, c = p & 15;
        float sm = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) sm += G[(4 * r + 6 + i) * GW + 4 * c + 6 + jj];
        SK[p] = sm * 0.0625f;
    }

    float amax = 0.f;        // MX: the largest activation this thread converted to an fp16 pair (range guard, bx_split.hpp)
    // ---- stage 2: conv2 4->8, s2 --------------------------------------------------------------
    {
        // conv1 inside conv2: a c2 pixel needs the 3 x 3 c1 pixels (2r + py, 2c + px), each a 3 x 3 window of the gray tile: 25 LDS reads
        // and 9 x 36 FMAs in registers instead of 36 reads of a c1 tile that first had to be computed, written (44 KB of LDS, the largest
        // tile of the kernel) and waited for behind a barrier.  2.25x the conv1 FLOPs (+ 12 % of the kernel's), one stage and 26 KB less.
        typedef float f2 __attribute__((ext_vector_type(2)));
        for (int e = tid; e < C2H * C2W; e += 512) {
            const int r = e / C2W, c = e - r * C2W;
            const int gy = 2 * Y4 - 2 + r, gx = 2 * X4 - 2 + c;
            float acc[8];
#pragma unroll
            for (int co = 0; co < 8; ++co) acc[co] = 0.f;
            if (gy >= 0 && gy < H2 && gx >= 0 && gx < W2) {
                float g[5][5];
#pragma unroll
                for (int i = 0; i < 5; ++i)
#pragma unroll
                    for (int j = 0; j < 5; ++j) g[i][j] = G[(2 * r + i) * GW + 2 * c + j];
                // c1[py][px][ch]: ReLU(conv1), zero outside the full-resolution map (= conv2's zero padding)
                f2 c1v[3][3][2];
#pragma unroll
                for (int py = 0; py < 3; ++py)
#pragma unroll
                    for (int px = 0; px < 3; ++px) {
                        f2 a01 = f2{bb1[0], bb1[1]}, a23 = f2{bb1[2], bb1[3]};
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx) {
                                const float* w = w1 + (dy * 3 + dx) * 4;
                                const f2 vv = f2{g[py + dy][px + dx], g[py + dy][px + dx]};
                                a01 = __builtin_elementwise_fma(vv, f2{w[0], w[1]}, a01);
                                a23 = __builtin_elementwise_fma(vv, f2{w[2], w[3]}, a23);
                            }
                        const int y1 = 4 * Y4 - 5 + 2 * r + py, x1 = 4 * X4 - 5 + 2 * c + px;
                        const bool ok = y1 >= 0 && y1 < H && x1 >= 0 && x1 < W;
                        // ReLU and the zero padding in ONE op per value: median(a, 0, hi) = max(a, 0) for hi = +inf, = 0 for hi = 0
                        const float hi = ok ? __builtin_inff() : 0.f;
                        c1v[py][px][0] = f2{__builtin_amdgcn_fmed3f(a01.x, 0.f, hi), __builtin_amdgcn_fmed3f(a01.y, 0.f, hi)};
                        c1v[py][px][1] = f2{__builtin_amdgcn_fmed3f(a23.x, 0.f, hi), __builtin_amdgcn_fmed3f(a23.y, 0.f, hi)};
                    }
                // conv2 on explicit 2-vectors: hipcc left the scalar form as 288 v_fmac_f32 per pixel (the PMC count of the kernel, 125 M vector
                // wave-instructions per launch = 93 % of its duration at 4 cycles each, says the kernel IS vector-issue bound: 45 % of them were this stage)
                f2 q[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) q[j] = f2{bb2[2 * j], bb2[2 * j + 1]};
#pragma unroll
                for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                    for (int py = 0; py < 3; ++py)
#pragma unroll
                        for (int px = 0; px < 3; ++px) {
                            const float v = ci & 1 ? c1v[py][px][ci >> 1].y : c1v[py][px][ci >> 1].x;
                            const float* w = w2 + ((ci * 9) + py * 3 + px) * 8;
                            const f2 vv = f2{v, v};
#pragma unroll
                            for (int j = 0; j < 4; ++j) q[j] = __builtin_elementwise_fma(vv, f2{w[2 * j], w[2 * j + 1]}, q[j]);
                        }
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[2 * j] = fmaxf(q[j].x, 0.f); acc[2 * j + 1] = fmaxf(q[j].y, 0.f); }
            }
            if constexpr (MX) {      // the pixel's 8 channels as fp16 pairs (even and odd columns of a row apart: block1_fx.hpp)
                uint4 h, l;
                split2_f16(acc[0], acc[1], h.x, l.x); split2_f16(acc[2], acc[3], h.y, l.y);
                split2_f16(acc[4], acc[5], h.z, l.z); split2_f16(acc[6], acc[7], h.w, l.w);
                amax = fmaxf(amax, fmaxf(fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])), fmaxf(fmaxf(acc[4], acc[5]), fmaxf(acc[6], acc[7]))));      // (ReLU'd: no fabs)
                unsigned char* p = reinterpret_cast<unsigned char*>(C2) + b1fx::c2_pixel_off(r, c);
                *reinterpret_cast<uint4*>(p) = h;
                *reinterpret_cast<uint4*>(p + b1fx::C2_PLANE) = l;
            } else {
#pragma unroll
                for (int co = 0; co < 8; ++co) C2[co * (C2H * C2W) + e] = acc[co];
            }
        }
    }
    if constexpr (MX) lds_dma_barrier();       // (conv3's weight image has landed, for every wave)
    else __syncthreads();
    if constexpr (MX) {      // the gray tile is dead: conv4's weight image takes its place (eleven 1-KiB pieces; it lands during stage 3)
        static_assert(b1fx::W4_BYTES <= (G_SZ + 3) * 4, "the weight image must fit into the gray tile");
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        for (int j = wv; j < b1fx::W4_BYTES / 1024; j += 8)
            __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const unsigned char*>(w4fx) + j * 1024 + (tid & 63) * 16),
                                             (lptr_t)(reinterpret_cast<unsigned char*>(lds) + j * 1024), 16, 0, 0);
    }

    // ---- stage 3 on the matrix cores: a column of the product = a pair of adjacent pixels (block1_fx.hpp); blocks of 16 consecutive pairs of the 17 x 17 pairs of
    // the tile; D: lane (pair ln, kg) holds couts 4 (kg & 1) + j of pixel 2 pc + (kg >> 1) ---------------------------------------------------------------------
    if constexpr (MX) {
        static_assert(b1fx::C2H == C2H && b1fx::C2W == C2W && 2 * b1fx::C2_PLANE == 8 * C2H * C2W * 4, "the fp16-pair planes fill the c2 tile exactly");
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        constexpr int NP = b1fx::NPAIR;
        const int lane = tid & 63, ln = lane & 15, kg = lane >> 4;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int dxw = kg - (ln >> 3);                                                                  // rows 8 .. 15: the right pixel, its window starts one column later
        const unsigned char* wa = reinterpret_cast<const unsigned char*>(lds + M_LDS_FLOATS) + (dxw >= 0 && dxw <= 2 ? b1fx::W3_REC * (8 * dxw + (ln & 7)) : b1fx::W3_ZERO_OFF);
        const float4 b3q = *reinterpret_cast<const float4*>(bb3 + 4 * (kg & 1));
        const float b3a[4] = {b3q.x, b3q.y, b3q.z, b3q.w};
        // lane constants of the two parity layouts: the c2 pixel (r + s, 2 pc + kg) sits at index pc + kgo of its row (column 35 -> index 34: zero weights and the dropped
        // pixel only), the output pixel 2 pc + (kg >> 1) at index pc + 17 (kg >> 1) of its c3 row
        const int kgo = (kg & 1) * b1fx::C2_NEVEN + (kg >> 1);
        const int c3o = 16 * (kg >> 1) * b1fx::C3_NEVEN + 8 * (kg & 1);
        auto blocks = [&](auto NBC, int blk0) __attribute__((always_inline)) {
            constexpr int NB = decltype(NBC)::value;
            int r[NB], pcs[NB];
            bool live[NB];
            const unsigned char* xb[NB];
            f32x4v acc[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int e0 = (blk0 + 8 * k) * 16 + ln, e = min(e0, C3H * NP - 1);
                const int rr = e / NP, pc = e - rr * NP;
                r[k] = rr; pcs[k] = pc;
                live[k] = e0 < C3H * NP && 2 * pc + (kg >> 1) < C3W;                                   // (the right pixel of a row's last pair does not exist)
                xb[k] = reinterpret_cast<const unsigned char*>(C2) + rr * b1fx::C2_ROWB + 16 * min(pc + kgo, C2W - 1);      // K step s: + s rows
                acc[k] = f32x4v{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int st = 0; st < 3; ++st) {
                const f16x8 q0 = *reinterpret_cast<const f16x8*>(wa + 32 * st), q2 = *reinterpret_cast<const f16x8*>(wa + 32 * st + 16);
                f16x8 q1 = q0 * (_Float16)0.00048828125f;            // fp16(w) = 2^-11 fp16(2^11 w): four v_pk_mul_f16 instead of a third fragment in LDS
                XFH_PIN(q1);                         // (computed HERE, in front of the MFMA group: left alone it sinks in between the MFMAs, into q2's registers)
                f16x8 xh[NB], xl[NB];
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    xh[k] = *reinterpret_cast<const f16x8*>(xb[k] + st * b1fx::C2_ROWB);
                    xl[k] = *reinterpret_cast<const f16x8*>(xb[k] + st * b1fx::C2_ROWB + b1fx::C2_PLANE);
                }
                // (q2, xh) (q1, xl) (q0, xh); the blocks take turns: a dependent MFMA waits for its predecessor to leave the pipe.  Vector code stays out of the group
                // and 16 idle slots behind it (a VALU result written right behind an MFMA can land in operand lanes the matrix core has not read yet: DESIGN 3.6,
                // tools/check_mfma_war.py); LDS reads may cross (mask 0x100): their results arrive long after
                __builtin_amdgcn_sched_barrier(0x100);
#pragma unroll
                for (int k = 0; k < NB; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(q2, xh[k], acc[k], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < NB; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(q1, xl[k], acc[k], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < NB; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(q0, xh[k], acc[k], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0x100);
                if constexpr (NB == 3) XFH_NOP16_3(acc[0], acc[1], acc[NB - 1]);
                else XFH_NOP16_2(acc[0], acc[NB - 1]);
                __builtin_amdgcn_sched_barrier(0x100);
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const int gy = 2 * Y4 - 1 + r[k], gx = 2 * X4 - 1 + 2 * pcs[k] + (kg >> 1);
                const float hi = gy >= 0 && gy < H2 && gx >= 0 && gx < W2 ? __builtin_inff() : 0.f;      // ReLU and the zero padding outside the map in one op (as conv1's)
                float y[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = __builtin_amdgcn_fmed3f(fmaf(acc[k][j], FX_SCALE_INV, b3a[j]), 0.f, hi);
                uint2 h, l;
                split2_f16(y[0], y[1], h.x, l.x); split2_f16(y[2], y[3], h.y, l.y);
                if (live[k]) {
                    amax = fmaxf(amax, fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3])));
                    unsigned char* p = reinterpret_cast<unsigned char*>(C3) + r[k] * b1fx::C3_ROWB + 16 * pcs[k] + c3o;
                    *reinterpret_cast<uint2*>(p) = h;
                    *reinterpret_cast<uint2*>(p + b1fx::C3_PLANE) = l;
                }
            }
        };
        // 19 blocks over 8 waves: wv, wv + 8 (, wv + 16 for waves 0 .. 2): two or three accumulator chains in flight
        if (wv + 16 < b1fx::NBLK3) blocks(std::integral_constant<int, 3>{}, wv);
        else blocks(std::integral_constant<int, 2>{}, wv);
    } else
    // ---- stage 3: conv3 8->8, s1 ----------------------------------
    for (int e = tid; e < C3H * C3W; e += 512) {
        const int r = e / C3W, c = e - r * C3W;
        const int gy = 2 * Y4 - 1 + r, gx = 2 * X4 - 1 + c;
        float acc[8];
#pragma unroll
        for (int co = 0; co < 8; ++co) acc[co] = 0.f;
        if (gy >= 0 && gy < H2 && gx >= 0 && gx < W2) {
#pragma unroll
            for (int co = 0; co < 8; ++co) acc[co] = bb3[co];
#pragma unroll 1
            for (int ci = 0; ci < 8; ++ci) {
                const float* src = C2 + ci * (C2H * C2W) + r * C2W + c;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float v = src[dy * C2W + dx];
                        const float* w = w3 + ((ci * 9) + dy * 3 + dx) * 8;
#pragma unroll
                        for (int co = 0; co < 8; ++co) acc[co] = fmaf(v, w[co], acc[co]);
                    }
            }
#pragma unroll
            for (int co = 0; co < 8; ++co) acc[co] = fmaxf(acc[co], 0.f);
        }
#pragma unroll
        for (int co = 0; co < 8; ++co) C3[co * (C3H * C3W) + e] = acc[co];
    }
    if constexpr (MX) {
        fx_report(amax, status);
        lds_dma_barrier();       // c3 is written and the weight image has landed, for every wave
    } else __syncthreads();

    // ---- stage 4 on the matrix cores: wave = output row, lane (ln, kg) = (output column, tap group); D: lane holds couts 16 cb + 4 kg + j of its pixel ----------
    if constexpr (MX) {
        static_assert(b1fx::C3H == C3H && b1fx::C3W == C3W && 2 * b1fx::C3_PLANE == 8 * C3H * C3W * 4, "the fp16-pair planes fill the c3 tile exactly");
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        const int lane = tid & 63, ln = lane & 15, kg = lane >> 4;
        const int orow = __builtin_amdgcn_readfirstlane(tid >> 6);
        const unsigned char* WI = reinterpret_cast<const unsigned char*>(lds);      // the weight image
        // one base per lane and weight region (block1_fx.hpp: A .. D; a lane without a real weight reads the zero block)
        const unsigned char* wA = WI + b1fx::A_OFF + 16 * lane;
        const unsigned char* wB = WI + (kg == 0 ? b1fx::B_OFF + b1fx::B_REC * ln : b1fx::ZERO_OFF);
        const unsigned char* wC = WI + (ln < 8 ? b1fx::C_OFF + b1fx::C_REC * (8 * kg + ln) : b1fx::ZERO_OFF);
        const unsigned char* wD = WI + (ln < 8 && kg == 0 ? b1fx::D_OFF + b1fx::D_REC * ln : b1fx::ZERO_OFF);
        auto wfrag = [&](int cb, int st, int q) {
            const unsigned char* base = cb == 0 ? (st < 2 ? wA : wB) : (st < 2 ? wC : wD);
            return *reinterpret_cast<const f16x8*>(base + b1fx::frag_in_rec(cb, st, q));
        };
        // the pixel under tap t = min(4 st + kg, 8) relative to pixel (2 orow, 2 ln): row t / 3, column parity / index of t % 3
        constexpr auto tap = [](int t) { return (t / 3) * b1fx::C3_ROWB + ((t % 3) == 1 ? 16 * b1fx::C3_NEVEN : 16 * ((t % 3) >> 1)); };
        const unsigned char* xb = reinterpret_cast<const unsigned char*>(C3) + b1fx::c3_pixel_off(2 * orow, 0) + 16 * ln;
        const int xo[3] = {kg == 0 ? tap(0) : kg == 1 ? tap(1) : kg == 2 ? tap(2) : tap(3), kg == 0 ? tap(4) : kg == 1 ? tap(5) : kg == 2 ? tap(6) : tap(7), tap(8)};
        f32x4v acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            const f16x8 xh = *reinterpret_cast<const f16x8*>(xb + xo[st]), xl = *reinterpret_cast<const f16x8*>(xb + xo[st] + b1fx::C3_PLANE);
            // (q2, xh) (q1, xl) (q0, xh), the two cout blocks taking turns: a dependent MFMA waits for its predecessor to leave the pipe.  Vector code stays out of
            // the group and 16 idle slots behind it (tools/check_mfma_war.py), LDS reads may cross
            f16x8 wq[2][3];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 3; ++q) wq[cb][q] = wfrag(cb, st, q);
            __builtin_amdgcn_sched_barrier(0x100);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[cb][2], xh, acc[cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[cb][1], xl, acc[cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[cb][0], xh, acc[cb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0x100);
            XFH_NOP16_2(acc[0], acc[1]);
            __builtin_amdgcn_sched_barrier(0x100);
        }
        // bias, ReLU, skip1 (1x1 conv of the 4 x 4 average) and the residual add; buffer stores: the lane's part of the address in ONE 32-bit register
        // (out of range = beyond the resource: dropped), the cout plane in the scalar offset
        const int oy = Y4 + orow, ox = X4 + ln;
        const float sk = sk_reg;
        const int plane = H4 * W4;
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(x1 + (size_t)b * 24 * plane), 0, 24 * plane * (int)sizeof(float), 0x00020000);
        const int voff = oy < H4 && ox < W4 ? (4 * kg * plane + oy * W4 + ox) * 4 : (int)0x80000000;
        const float* bp = bb4 + 4 * kg;
        const float* wp = skw + 4 * kg;
        const float* sp = skb + 4 * kg;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const bool on = cb == 0 || kg < 2;             // couts 24 .. 31 do not exist
            const float4 bq = on ? *reinterpret_cast<const float4*>(bp + 16 * cb) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 wq = on ? *reinterpret_cast<const float4*>(wp + 16 * cb) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 sq = on ? *reinterpret_cast<const float4*>(sp + 16 * cb) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float bqa[4] = {bq.x, bq.y, bq.z, bq.w}, wqa[4] = {wq.x, wq.y, wq.z, wq.w}, sqa[4] = {sq.x, sq.y, sq.z, sq.w};
            const int vo = on ? voff : (int)0x80000000;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fmaxf(fmaf(acc[cb][j], FX_SCALE_INV, bqa[j]), 0.f) + fmaf(sk, wqa[j], sqa[j])), rs_out, vo,
                                                      (16 * cb + j) * plane * 4, 0);
        }
    } else
    // ---- stage 4: conv4 8->24, s2 + skip1 + residual add; thread = (pixel, 6 of 24 couts) -------
    {
        const int p = tid & 127, r = p >> 4, c = p & 15;
        const int g = __builtin_amdgcn_readfirstlane(tid >> 7);      // wave pair -> couts 6g .. 6g+5
        const int oy = Y4 + r, ox = X4 + c;
        float acc[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[j] = bb4[g * 6 + j];
#pragma unroll 1
        for (int ci = 0; ci < 8; ++ci) {
            const float* src = C3 + ci * (C3H * C3W) + (2 * r) * C3W + 2 * c;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float v = src[dy * C3W + dx];
                    const float* w = w4 + ((ci * 9) + dy * 3 + dx) * 24 + g * 6;
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc[j] = fmaf(v, w[j], acc[j]);
                }
        }
        // skip1: 4x4 average of the gray tile (AvgPool2d(4,4), computed once in stage 1), then 1x1 conv 1->24 with bias
        const float sk = SK[p];
        if (oy < H4 && ox < W4) {
            float* op = x1 + (((size_t)b * 24 + g * 6) * H4 + oy) * W4 + ox;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float v = fmaxf(acc[j], 0.f) + fmaf(sk, skw[g * 6 + j], skb[g * 6 + j]);
                op[(size_t)j * H4 * W4] = v;
            }
        }
    }
}

}  // namespace xfh
