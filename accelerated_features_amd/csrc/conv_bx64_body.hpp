// conv_bx64_kernel (k_conv_bx64.hip): the kernel body in a header of its own so that tests/emu/ can compile the SAME source for the host (XFH_HOST_EMU) and run it
// against a float64 convolution without a GPU.
#pragma once
#ifndef XFH_HOST_EMU
#include "kernels.hpp"
#include <type_traits>
#ifndef XFH_DYN_LDS_BYTES
#define XFH_DYN_LDS_BYTES(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif
#define XFH_LDS_ADDR(p, base) ((unsigned)(size_t)(__attribute__((address_space(3))) void*)(p))
/* LDS-DMA of 16 bytes per lane: M0 = LDS address of the 1-KiB piece, the lane's part of the global address in voff (inline asm: hipcc would make every LDS read wait for all DMA it can see) */
#define XFH_DMA_B128_TO_LDS(m0v, voff, rsrc, soff) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane((int)(m0v))), "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane((int)(soff))) : "memory")      /* (readfirstlane: both are wave-uniform by construction; where hipcc cannot see it, it hands the asm a vector register) */
#define XFH_WAIT_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define XFH_NOP16() asm volatile("s_nop 7\n\ts_nop 7")
#ifndef XFH_NOP16_2
#define XFH_NOP16_2(a, b) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b))
#endif
#define XFH_NOP16_4(a, b, c, d) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define XFH_NOP32_2(a, b) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b))
#ifndef XFH_GPTR_DEFINED
#define XFH_GPTR_DEFINED
typedef __attribute__((address_space(1))) const void* xfh_gptr_t;
typedef __attribute__((address_space(3))) void* xfh_lptr_t;
#endif
#endif
#include "bx_split.hpp"

namespace xfh {

typedef int i32x4 __attribute__((ext_vector_type(4)));

struct Bx64Args {
    const float* in;
    const void* wq;            // [cin/16][3 dy][3 dx][2 cout blocks][3 splits][64 lanes][8 bf16]   (api.hip)
    const float* bias;
    float* out;
    int relu, H, W, B;
    int ncols, nhr, upi;       // 16-column strips, 8-row half tiles per strip, units per image
    long long* trace;
    // fused trailing 1x1 (64 -> 64): split weights in the K order of the 3x3's D registers, [K step 4][cout block 2][split 3][64 lanes] 8 bf16
    const uint4* wq2;
    const float* bias2;
    int relu2;
    int cold;
    int* status;               // fx: range guard (bx_split.hpp), may be NULL
    const void* zeros;         // SP input: >= 16 bytes of zeros (the halo outside the map is DMA'd from there)
};

namespace bx64 {
constexpr int XROWB = 2048, SPLB = 32, IW = 18, IH = 18;
// bytes per staged pixel: 16 channels x (3 bf16 | 2 fp16 fragments) + 16: an ODD multiple of 16 B keeps the 16 lanes of a ds_read_b128 group on distinct banks
template <bool FX> constexpr int pixb() { return FX ? 80 : 112; }
constexpr int X_BYTES = IH * XROWB;                    // 36864
constexpr int STEP_BYTES = 2 * 3 * 1024, SLOT_BYTES = 3 * STEP_BYTES, NPIECE = SLOT_BYTES / 1024;      // 6 KiB per K step, 18 per slot
constexpr int RING_OFF = X_BYTES, BIAS_OFF = RING_OFF + 2 * SLOT_BYTES, LDS_BYTES = BIAS_OFF + 128 * 4;      // bias of the 3x3, bias of the fused 1x1
// SP ("split, producer-side"): the input arrives as fp16 pairs written by the layer before -- per image [16-channel chunk][pixel][64 bytes: hi ch 0-7 | hi 8-15 | lo 0-7 | lo 8-15]
// (4 bytes per value: the HBM traffic of fp32) -- and is staged by LDS-DMA alone: no raw values in registers, no split, no ds_write, no staging phase between barriers.
// LDS tile of a chunk: [18 rows][18 pixels][64 bytes], two of them (chunk c + 1 lands while chunk c is multiplied); the 16-byte slot of a pixel sits at
// slot ^ ((column >> 2) & 3): with the 64-byte pitch that keeps the 16 lanes of a ds_read_b128 group on distinct banks (unit 4 (x & 3) + (slot ^ (x >> 2)) mod 16) without padding.
constexpr int XSP_ROWB = IW * 64, XSP_BYTES = IH * XSP_ROWB, XSP_UNITS = XSP_BYTES / 16, XSP_NDMA = (XSP_BYTES + 1023) / 1024;      // 1152, 20736, 1296, 21
constexpr int SP_RING_OFF = 2 * XSP_BYTES, SP_BIAS_OFF = SP_RING_OFF + 2 * SLOT_BYTES, SP_LDS_BYTES = SP_BIAS_OFF + 128 * 4;                 // 78848: two workgroups per CU
static_assert(2 * SP_LDS_BYTES <= 160 * 1024, "two workgroups per CU");
constexpr int NQ = 6;                                   // aligned 4-pixel quads per halo row
static_assert(IH * NQ * 2 <= 256, "one (row, quad, 8-channel group) item per thread");
}

// FUSE: 0 = the 3x3 alone; 1 = + trailing 1x1 (64 -> 64), NCHW output; 2 = the same with channels-last output
// FX: the fp16-pair arithmetic (bx_split.hpp) -- two input fragments per pixel, three MFMAs per K step and accumulator instead of six
// FXM: 0 = bf16 three-way split, 1 = fp16 pair, 2 = fp16 pair with TWO weight fragments per (tap, cout block) in the stream and in LDS (q0, q2; q1 = fp16(w) = 2^-11 q0
// derived with four v_pk_mul_f16 per cout block and K step): a third less weight DMA, 8 instead of 10 LDS reads per 12 MFMAs
// SP: bit 1 = the input is in the split format above (a.in), bit 2 = the output is written in it (FUSE 0 only) -- both need the fp16-pair arithmetic
template <int CIN, int FUSE, int FXM, int SP = 0>
__device__ __forceinline__ void conv_bx64_body(const Bx64Args& a) {
    using namespace bx64;
    constexpr bool FX = FXM > 0;
    constexpr bool IN_SP = (SP & 1) != 0, OUT_SP = (SP & 2) != 0;
    static_assert(!SP || (FX && CIN == 64), "the split format is the fp16 pair's");
    static_assert(!OUT_SP || FUSE == 0, "only the plain 3x3 writes the split format");
    constexpr int ROFF = IN_SP ? SP_RING_OFF : RING_OFF, BOFF = IN_SP ? SP_BIAS_OFF : BIAS_OFF;
    constexpr int NWF = FXM == 2 ? 2 : 3;
    constexpr int STEP_B = 2 * NWF * 1024, SLOT_B = 3 * STEP_B, NPC = SLOT_B / 1024;      // (the ring keeps the room of the three-fragment form: bx64::LDS_BYTES)
    constexpr int PIXB = pixb<FX>(), NXS = FX ? 2 : 3;
    using frag_t = std::conditional_t<FX, f16x8, bf16x8>;
    auto mfma = [](frag_t x, frag_t y, f32x16 c) __attribute__((always_inline)) {
        if constexpr (FX) return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
    };
    constexpr int NCH = CIN / 16, NROW = NCH * 3, COUT = 64;
    static_assert(NROW % 2 == 0, "the ring slot of a row must not depend on the tile");
    XFH_DYN_LDS_BYTES(smem_b64);
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t HW = (size_t)a.H * a.W;
    float* bias_lds = reinterpret_cast<float*>(smem_b64 + BOFF);
    if (tid < 64) bias_lds[tid] = a.bias[tid];
    if (FUSE && tid >= 64 && tid < 128) bias_lds[tid] = a.bias2[tid - 64];

    // ---- this workgroup's units -----------------------------------------------------------------------------------------
    // unit u of an image list = (image, 16-column strip, half-tile row), strips and rows fastest.  With a batch that is a multiple of
    // 8 the images of XCD x are x, x + 8, ... (workgroup id & 7 = XCD): a strip's neighbours share an L2.
    int u0, u1, img0, img_step;
    {
        const int G = (int)gridDim.x, g = (int)blockIdx.x;
        if (xcd_swizzled(a.B) && (G & 7) == 0) {
            const long long U = (long long)(a.B >> 3) * a.upi;
            const int slot = g >> 3, nslot = G >> 3;
            u0 = (int)(U * slot / nslot); u1 = (int)(U * (slot + 1) / nslot);
            img0 = g & 7; img_step = 8;
        } else {
            const long long U = (long long)a.B * a.upi;
            u0 = (int)(U * g / G); u1 = (int)(U * (g + 1) / G);
            img0 = 0; img_step = 1;
        }
    }
    if (u0 >= u1) return;
    struct Tile { int b, y0, x0, full; };
    auto tile_at = [&](int u, Tile& t) {      // returns the units consumed (2 = a full 16-row tile)
        const int im = u / a.upi, rem = u - im * a.upi;
        const int col = rem / a.nhr, hr = rem - col * a.nhr;
        t.b = img0 + img_step * im; t.y0 = hr * 8; t.x0 = col * 16;
        t.full = (u + 1 < u1 && hr + 1 < a.nhr) ? 1 : 0;
        return 1 + t.full;
    };

    // ---- LDS-DMA of the weight stream (inline asm: hipcc would make every LDS read wait for all DMA it can see) -------------
    auto make_rsrc = [](const void* p, unsigned bytes) {
        const unsigned long long ba = (unsigned long long)p;
        i32x4 r;
        r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ba);
        r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(ba >> 32) & 0xffffu));
        r.z = __builtin_amdgcn_readfirstlane((int)bytes);
        r.w = 0x00020000;
        return r;
    };
    const i32x4 rs_w = make_rsrc(a.wq, (unsigned)(NROW * SLOT_B));
    const int dma_voff = lane * 16;
    auto lds_addr = [&](const unsigned char* p) { return XFH_LDS_ADDR(p, smem_b64); };
    auto issue_row = [&](int r) __attribute__((always_inline)) {             // weights of row r (chunk r / 3, tap row r % 3) -> slot r & 1
        for (int j = wave; j < NPC; j += 4) {
            const unsigned m0v = lds_addr(smem_b64 + ROFF + (r & 1) * SLOT_B + j * 1024);
            const int soff = r * SLOT_B + j * 1024;
            XFH_DMA_B128_TO_LDS(m0v, dma_voff, rs_w, soff);
        }
    };
    auto dma_barrier = [&]() {                // everything this workgroup has in flight has landed, for every wave
        XFH_WAIT_VMCNT0();
        __syncthreads();
    };

    // ---- raw fp32 values of one 16-channel chunk of a tile.  Item of a thread = 4 consecutive pixels x 8 channels: eight
    // buffer_load_dwordx4 (one per channel plane) instead of 32 dword loads -- the texture addresser takes ~16 cycles per wave
    // instruction whatever its width, and eight waves loading dword by dword kept it busy for 3 k cycles per chunk.  The halo row
    // [x0 - 1, x0 + 17) is covered by the six aligned quads [x0 - 4, x0 + 20); W % 4 == 0 keeps every quad entirely inside or outside.
    const bool has_item = tid < IH * NQ * 2;
    const int it_g8 = tid / (IH * NQ), it_row = (tid - it_g8 * (IH * NQ)) / NQ, it_quad = tid % NQ;
    float v[8][4];
    int v_gx = 0;                             // first column of the quad in flight (the tail of a quad that straddles the right border is masked
                                              // when it is consumed: touching the values where they are loaded would park a vmcnt(0) there)
    auto issue_loads = [&](const Tile& t, int chunk) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)t.b * CIN * HW), 0, (int)(CIN * HW * sizeof(float)), 0x00020000);
        const int nrow = t.full ? 18 : 10;
        const int gy = t.y0 - 1 + it_row, gx = t.x0 - 4 + 4 * it_quad;
        const bool ok = has_item && it_row < nrow && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        v_gx = gx;
        const int voff = ok ? (int)((((size_t)it_g8 * 8) * HW + (size_t)gy * a.W + gx) * 4) : (int)0x80000000;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const auto q = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, (int)((chunk * 16 + k) * HW * 4), 0);
            v[k][0] = __uint_as_float(q[0]); v[k][1] = __uint_as_float(q[1]); v[k][2] = __uint_as_float(q[2]); v[k][3] = __uint_as_float(q[3]);
        }
    };
    // split3 works on the two neighbouring PIXELS of a loaded quad (adjacent registers of one dwordx4: pairing channels instead made
    // hipcc re-arrange all 32 values with moves right behind the loads -- and wait for them there); v_perm_b32 then gathers the
    // channel pairs of each pixel: 16 + 16 bits from two registers in one op.
    auto stage_write = [&]() __attribute__((always_inline)) {
        if (!has_item) return;
        unsigned amax = 0;                        // fx: the largest fp16 high parts of this item (range guard, bx_split.hpp; a kernel-long register cost the fused forms eight spills)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            unsigned H[8], M[8], L[8];                     // {pixel 2 pp, pixel 2 pp + 1} of channel k
            // beyond the right border (W % 4 != 0 only) the quad's tail holds the next row's first pixels: zero them as values, not in v
            // (conditional stores into the array sent it to scratch memory)
            const bool z0 = (a.W & 3) && v_gx + 2 * pp >= a.W, z1 = (a.W & 3) && v_gx + 2 * pp + 1 >= a.W;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float x0 = v[k][2 * pp], x1 = v[k][2 * pp + 1];
                if (a.W & 3) { x0 = z0 ? 0.f : x0; x1 = z1 ? 0.f : x1; }
                if constexpr (FX) { split2_f16(x0, x1, H[k], L[k]); fx_track_h(amax, H[k], true); M[k] = 0; }
                else split3(x0, x1, H[k], M[k], L[k]);
            }
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                const int cc = 4 * it_quad + 2 * pp + e2 - 3;          // halo column of this pixel
                if (cc < 0 || cc >= IW) continue;                     // (quad 0: its last pixel only; quad 5: its first only)
                const unsigned sel = e2 ? 0x07060302u : 0x05040100u;
                uint4 h, m, l;
                h.x = __builtin_amdgcn_perm(H[1], H[0], sel); h.y = __builtin_amdgcn_perm(H[3], H[2], sel);
                h.z = __builtin_amdgcn_perm(H[5], H[4], sel); h.w = __builtin_amdgcn_perm(H[7], H[6], sel);
                m.x = __builtin_amdgcn_perm(M[1], M[0], sel); m.y = __builtin_amdgcn_perm(M[3], M[2], sel);
                m.z = __builtin_amdgcn_perm(M[5], M[4], sel); m.w = __builtin_amdgcn_perm(M[7], M[6], sel);
                l.x = __builtin_amdgcn_perm(L[1], L[0], sel); l.y = __builtin_amdgcn_perm(L[3], L[2], sel);
                l.z = __builtin_amdgcn_perm(L[5], L[4], sel); l.w = __builtin_amdgcn_perm(L[7], L[6], sel);
                unsigned char* p = smem_b64 + it_row * XROWB + cc * PIXB + it_g8 * 16;
                *reinterpret_cast<uint4*>(p) = h;
                if constexpr (FX) *reinterpret_cast<uint4*>(p + SPLB) = l;
                else {
                    *reinterpret_cast<uint4*>(p + SPLB) = m;
                    *reinterpret_cast<uint4*>(p + 2 * SPLB) = l;
                }
            }
        }
        if constexpr (FX) fx_report_h(amax, a.status);
    };

    // ---- SP input: the chunk tiles by LDS-DMA.  DMA instruction i of a chunk fills LDS units [64 i, 64 i + 64) of the tile (16 bytes per lane); wave w issues i = w, w + 4, ...
    // Per lane and instruction the (row, column, logical slot) of its unit are constants of the kernel; per tile and chunk they give the source address (or the zeros).
    int sp_it[(XSP_NDMA + 3) / 4];
    if constexpr (IN_SP) {
#pragma unroll
        for (int k = 0; k < (XSP_NDMA + 3) / 4; ++k) {
            const int i = wave + 4 * k, u = i * 64 + lane;
            const int row = u / (IW * 4), rem = u - row * (IW * 4), px = rem >> 2, logical = (rem & 3) ^ ((px >> 2) & 3);
            sp_it[k] = i < XSP_NDMA && u < XSP_UNITS ? (row << 16) | (px << 8) | logical : -1;
        }
    }
    auto issue_chunk = [&](const Tile& t, int chunk, int buf) __attribute__((always_inline)) {
        if constexpr (IN_SP) {
            const int nrow = t.full ? 18 : 10;
            const unsigned char* src = reinterpret_cast<const unsigned char*>(a.in) + ((size_t)t.b * NCH + chunk) * HW * 64;
#pragma unroll
            for (int k = 0; k < (XSP_NDMA + 3) / 4; ++k) {
                const int it = sp_it[k], row = it >> 16, px = (it >> 8) & 0xff, logical = it & 0xff;
                if (it < 0 || row >= nrow) continue;                      // (wave-uniform only for whole instructions; partly live ones run under the exec mask)
                const int gy = t.y0 - 1 + row, gx = t.x0 - 1 + px;
                const bool in = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                const unsigned char* g = in ? src + ((size_t)gy * a.W + gx) * 64 + logical * 16 : reinterpret_cast<const unsigned char*>(a.zeros);
                __builtin_amdgcn_global_load_lds((xfh_gptr_t)g, (xfh_lptr_t)(smem_b64 + buf * XSP_BYTES + (wave + 4 * k) * 1024), 16, 0, 0);
            }
        }
    };
    // the lane's read offsets: pixel (row l31 >> 4, column (l31 & 15) + s) of the block, part q (0 = high, 1 = low), channel half `half`: slot (2 q + half) ^ ((x >> 2) & 3)
    int xsp[3][2];
    if constexpr (IN_SP) {
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            const int x = (l31 & 15) + s3;
#pragma unroll
            for (int q = 0; q < 2; ++q) xsp[s3][q] = (l31 >> 4) * XSP_ROWB + x * 64 + (((2 * q + half) ^ ((x >> 2) & 3)) << 4);
        }
    }

    long long* tr = a.trace && tid == 0 ? a.trace + (size_t)blockIdx.x * 64 : nullptr;
    int tix = 0;
#define BX_STAMP(k) { if (tr && tix == 1) tr[k] = __builtin_amdgcn_s_memtime(); }      /* second tile of the workgroup: [0] start, per row r: [1+4r] staged / row start, [2+4r] barrier passed, [3+4r] DMA + loads issued, [4+4r] MFMAs issued; [50] stores issued, [51] end barrier */
    struct Frag { frag_t x[2][NXS]; frag_t w[2][3]; };
    const int lane_px = (l31 >> 4) * XROWB + (l31 & 15) * PIXB + half * 16;

    // ---- one tile: NPB pixel blocks per wave (2 = 16x16 tile, 1 = 8x16 half tile).  Two instantiations of the whole tile body:
    // accumulators that live across a branch between two tap-row variants were given a second register set and 64 moves per row.
    auto do_tile = [&](auto NPBC, const Tile& cur, const Tile& nxt, bool has_next) __attribute__((always_inline)) {
        constexpr int NPB = decltype(NPBC)::value;
        f32x16 acc[NPB][2];                   // [pixel block][cout block]
#pragma unroll
        for (int j = 0; j < NPB; ++j)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][cb][r] = 0.f;
        // block rows of this wave: full tile 2 w, 2 w + 1 ; half tile w
        const int br0 = NPB == 2 ? 2 * wave : wave, br1 = 2 * wave + 1;
        const int xb[2] = {2 * br0 * XROWB + lane_px, 2 * br1 * XROWB + lane_px};
        for (int c = 0; c < NCH; ++c) {
            if constexpr (!IN_SP) {
                if (c > 0) dma_barrier();      // every wave has finished the previous chunk's last tap row (the tile loop ends on a barrier)
                stage_write();
            }
            for (int dy = 0; dy < 3; ++dy) {
                const int r = c * 3 + dy;
                BX_STAMP(1 + 4 * r)
                dma_barrier();
                BX_STAMP(2 + 4 * r)                 // row r's weights landed; (dy = 0) the chunk is staged; (dy > 0) row r - 1 is finished
                BX_STAMP(3 + 4 * r)
                // ---- one tap row: 3 K steps x (NPB pixel blocks x 2 cout blocks) x 6 MFMAs; operands of step s+1 read under step s
                const unsigned char* wslot = smem_b64 + ROFF + (r & 1) * SLOT_B + lane * 16;
                const unsigned char* xrow = smem_b64 + (IN_SP ? (c & 1) * XSP_BYTES + dy * XSP_ROWB : dy * XROWB);
                Frag f[2];
                auto load = [&](int s, Frag& o) {
#pragma unroll
                    for (int j = 0; j < NPB; ++j)
#pragma unroll
                        for (int q = 0; q < NXS; ++q)
                            o.x[j][q] = IN_SP ? *reinterpret_cast<const frag_t*>(xrow + 2 * (j ? br1 : br0) * XSP_ROWB + xsp[s][q < 2 ? q : 0])
                                              : *reinterpret_cast<const frag_t*>(xrow + xb[j] + s * PIXB + q * SPLB);
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int q = 0; q < NWF; ++q) o.w[cb][NWF == 3 ? q : 2 * q] = *reinterpret_cast<const frag_t*>(wslot + s * STEP_B + (cb * NWF + q) * 1024);
                };
                load(0, f[0]);
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    if constexpr (NWF == 2) {      // q1 from q0 (the registers it lands in were last read two K steps ago)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb) f[s & 1].w[cb][1] = f[s & 1].w[cb][0] * (_Float16)0.00048828125f;
                    }
                    const Frag& cf = f[s & 1];
                    if (s + 1 < 3) load(s + 1, f[(s + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    // products (weight split, input split), small terms first: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h); independent accumulators
#define BX_MM(WQ, XQ) { _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) _Pragma("unroll") for (int j = 0; j < NPB; ++j) \
                        acc[j][cb] = mfma(cf.w[cb][WQ], cf.x[j][XQ], acc[j][cb]); }
                    if constexpr (FX) { BX_MM(2, 0) BX_MM(1, 1) BX_MM(0, 0) }      // fp16 pair: (2^11 w - q0) xh, w xl, q0 xh -- all at scale 2^11
                    else { BX_MM(2, 0) BX_MM(0, 2) BX_MM(1, 1) BX_MM(1, 0) BX_MM(0, 1) BX_MM(0, 0) }
#undef BX_MM
                    __builtin_amdgcn_sched_barrier(0);
                    // memory instructions go BETWEEN the MFMA groups: their issue (~100 cycles per LDS-DMA piece or load with the CU's
                    // eight waves at it) overlaps the matrix pipe's backlog instead of preceding it
                    // (and idle slots first: VALU address arithmetic right behind an MFMA may land in operand lanes it has not read yet)
                    if (s < 2) { XFH_NOP16(); __builtin_amdgcn_sched_barrier(0); }
                    if (s == 0) issue_row(r + 1 < NROW ? r + 1 : 0);          // next row (of the next tile after the last one: the stream is cyclic)
                    if (s == 1 && dy == 0) {   // raw values of the next chunk (or the next tile's first) fly under this chunk's MFMAs
                        // (ONE load site: two sites load into two register sets and merge them with moves -- behind a wait for the loads)
                        const bool same = c + 1 < NCH;
                        Tile lt;
                        lt.b = same ? cur.b : nxt.b; lt.y0 = same ? cur.y0 : nxt.y0; lt.x0 = same ? cur.x0 : nxt.x0; lt.full = same ? cur.full : nxt.full;
                        if (same || has_next) { if constexpr (IN_SP) issue_chunk(lt, same ? c + 1 : 0, (c + 1) & 1); else issue_loads(lt, same ? c + 1 : 0); }
                    }
                }
                if constexpr (NPB == 2) XFH_NOP16_4(acc[0][0], acc[0][1], acc[1][0], acc[1][1]);      // (tied to the accumulators: an asm
                else XFH_NOP16_2(acc[0][0], acc[0][1]);                                                            // without operands is no anchor)
                __builtin_amdgcn_sched_barrier(0);
                BX_STAMP(4 + 4 * r)
            }
        }
        if constexpr (FUSE == 0 && OUT_SP) {
            // ---- bias, ReLU, and the output as fp16 pairs in the split format of the next layer: the lane's four channels of a register quad g4 (cout block cb) are
            // channels 8 (g4 & 1) + 4 half + e of chunk 2 cb + (g4 >> 1): 8 bytes into the high slot g4 & 1, 8 bytes into the low slot (+ 32) of the pixel's record
            typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<unsigned char*>(a.out) + (size_t)cur.b * NCH * HW * 64), 0, (int)(NCH * HW * 64), 0x00020000);
            const int ox = cur.x0 + (l31 & 15);
            unsigned amaxo = 0;                   // range guard of what the next layer will multiply (on the high parts: bx_split.hpp)
#pragma unroll
            for (int j = 0; j < NPB; ++j) {
                const int oy = cur.y0 + 2 * (j ? br1 : br0) + (l31 >> 4);
                const int voff = oy < a.H && ox < a.W ? (oy * a.W + ox) * 64 + half * 8 : (int)0x80000000;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const float4 t = *reinterpret_cast<const float4*>(bias_lds + cb * 32 + 8 * g4 + 4 * half);
                        const float bq[4] = {t.x, t.y, t.z, t.w};
                        float y[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            y[e] = fmaf(acc[j][cb][4 * g4 + e], FX_SCALE_INV, bq[e]);
                            if (a.relu) y[e] = fmaxf(y[e], 0.f);
                        }
                        u32x2v h, l;
                        unsigned h0, l0, h1, l1;
                        split2_f16(y[0], y[1], h0, l0); split2_f16(y[2], y[3], h1, l1);
                        fx_track_h(amaxo, h0, true); fx_track_h(amaxo, h1, true);
                        h[0] = h0; h[1] = h1; l[0] = l0; l[1] = l1;
                        const int soff = (2 * cb + (g4 >> 1)) * (int)HW * 64 + (g4 & 1) * 16;
                        __builtin_amdgcn_raw_buffer_store_b64(h, rs_out, voff, soff, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(l, rs_out, voff, soff + 32, 0);
                    }
            }
            fx_report_h(amaxo, a.status);
        } else if constexpr (FUSE == 0) {
            // ---- bias, ReLU, buffer stores (lanes outside the image carry an out-of-range offset).  A = weights, B = pixels: lane (pixel,
            // half) holds couts (r & 3) + 8 (r >> 2) + 4 half; a store instruction writes four 64-byte row segments.  (The transposed
            // product -- lane = cout, four consecutive pixels per register quad, dwordx4 stores -- has a quarter of the instructions but
            // every lane in its own cache line: 64 lines per instruction instead of 4, and was slower: the addresser works per line.)
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)cur.b * COUT * HW), 0, (int)(COUT * HW * sizeof(float)), 0x00020000);
            const int ox = cur.x0 + (l31 & 15);
    #pragma unroll
            for (int j = 0; j < NPB; ++j) {
                const int oy = cur.y0 + 2 * (j ? br1 : br0) + (l31 >> 4);
                const int voff = oy < a.H && ox < a.W ? (int)(((size_t)(4 * half) * HW + (size_t)oy * a.W + ox) * 4) : (int)0x80000000;
    #pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    float bs[16];
    #pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const float4 t = *reinterpret_cast<const float4*>(bias_lds + cb * 32 + 8 * g4 + 4 * half);
                        bs[4 * g4] = t.x; bs[4 * g4 + 1] = t.y; bs[4 * g4 + 2] = t.z; bs[4 * g4 + 3] = t.w;
                    }
    #pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float y = FX ? fmaf(acc[j][cb][r], FX_SCALE_INV, bs[r]) : acc[j][cb][r] + bs[r];
                        if (a.relu) y = fmaxf(y, 0.f);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rs_out, voff, (int)((cb * 32 + (r & 3) + 8 * (r >> 2)) * HW * 4), 0);
                    }
                }
            }
        } else {
            // ---- fused trailing 1x1 (block3.2 / block_fusion.2) on the same matrix cores: the 3x3's D registers (lane = pixel, registers =
            // couts (r & 3) + 8 (r >> 2) + 4 half), biased and ReLU'd, ARE the 1x1's pixel-side fragments once split: K step t of lane half h
            // takes the register quads 8 (t & 1), 8 (t & 1) + 4 of cout block t >> 1 (the weights are packed in that K order, as for the
            // heads' chained layers).  Weight fragments come straight from L2 (24 KiB, the same for every wave; no LDS left for them),
            // one K step per load batch; the split fragments are double-buffered and kept alive as in head_bx_layer (MFMA operand hazard).
            float amax2 = 0.f;                    // fx: range guard of the 1x1's input (on the values, not on the split's high parts as the staging does: + 1 register = 8 bytes of scratch in FUSE 2)
#pragma unroll
            for (int j = 0; j < NPB; ++j)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const float4 t = *reinterpret_cast<const float4*>(bias_lds + cb * 32 + 8 * g4 + 4 * half);
                        const float bq[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float y = FX ? fmaf(acc[j][cb][4 * g4 + e], FX_SCALE_INV, bq[e]) : acc[j][cb][4 * g4 + e] + bq[e];
                            if (a.relu) y = fmaxf(y, 0.f);
                            if constexpr (FX) amax2 = fmaxf(amax2, fabsf(y));
                            acc[j][cb][4 * g4 + e] = y;
                        }
                    }
            if constexpr (FX) fx_report(amax2, a.status);
            // one pixel block at a time (its two 1x1 accumulators, stores included): both blocks at once do not fit into 256 registers next to
            // the 3x3's results and the next tile's prefetched input
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)cur.b * 64 * HW), 0, (int)(64 * HW * sizeof(float)), 0x00020000);
            frag_t w2[2][3], xf[2][NXS];
            // (buffer loads: ONE address register per lane, the fragment in the scalar offset -- as global loads the 24 fragment addresses were
            // 48 registers, spilled, and re-read from scratch in front of every load)
            const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.wq2, 0, 4 * 2 * 3 * 1024, 0x00020000);
            auto ldw2 = [&](int t) {
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                    for (int q = 0; q < 3; ++q) w2[m2][q] = __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, lane * 16, ((t * 2 + m2) * 3 + q) * 1024, 0));
            };
#pragma unroll
            for (int j = 0; j < NPB; ++j) {
                f32x16 acc2[2];                   // [cout block of the 1x1]
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                    for (int r = 0; r < 16; ++r)      // FUSE 1: D2 rows = couts ; FUSE 2 (transposed product): D2 columns = couts
                        acc2[m2][r] = (FUSE == 1 ? bias_lds[64 + m2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] : bias_lds[64 + m2 * 32 + l31]) * (FX ? 2048.f : 1.f);      // (fx: the accumulator lives at scale 2^11)
                auto split_step = [&](int t, frag_t (&o)[NXS]) {
                    uint4 uh, um, ul;
                    unsigned* ph = &uh.x; unsigned* pm = &um.x; unsigned* pl = &ul.x;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if constexpr (FX) { split2_f16(acc[j][t >> 1][8 * (t & 1) + 2 * i], acc[j][t >> 1][8 * (t & 1) + 2 * i + 1], ph[i], pl[i]); }
                        else split3(acc[j][t >> 1][8 * (t & 1) + 2 * i], acc[j][t >> 1][8 * (t & 1) + 2 * i + 1], ph[i], pm[i], pl[i]);
                    }
                    o[0] = __builtin_bit_cast(frag_t, uh);
                    if constexpr (FX) o[1] = __builtin_bit_cast(frag_t, ul);
                    else { o[1] = __builtin_bit_cast(frag_t, um); o[2] = __builtin_bit_cast(frag_t, ul); }
                };
                asm volatile("" ::: "memory");
                ldw2(0);
                split_step(0, xf[0]);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int sb = t & 1;
                    __builtin_amdgcn_sched_barrier(0);
                    // products (weight split, input split), small terms first; FUSE 2 swaps the operands (rows = pixels, lane = cout)
#define BX_MM2(WQ, XQ) { _Pragma("unroll") for (int m2 = 0; m2 < 2; ++m2) acc2[m2] = FUSE == 1 \
                        ? mfma(w2[m2][WQ], xf[sb][XQ], acc2[m2]) : mfma(xf[sb][XQ], w2[m2][WQ], acc2[m2]); }
                    if constexpr (FX) { BX_MM2(2, 0) BX_MM2(1, 1) BX_MM2(0, 0) }
                    else { BX_MM2(2, 0) BX_MM2(0, 2) BX_MM2(1, 1) BX_MM2(1, 0) BX_MM2(0, 1) BX_MM2(0, 0) }
#undef BX_MM2
                    __builtin_amdgcn_sched_barrier(0);
                    if (t + 1 < 4) {
                        split_step(t + 1, xf[sb ^ 1]);
                        // the new fragments pass through an asm that uses the old ones and this step's weights: their registers stay occupied
                        // while the split's results and temporaries are written
#ifndef XFH_HOST_EMU
                        asm volatile("" : "+v"(xf[sb ^ 1][0]), "+v"(xf[sb ^ 1][1]), "+v"(xf[sb ^ 1][NXS - 1])
                                        : "v"(xf[sb][0]), "v"(xf[sb][1]), "v"(xf[sb][NXS - 1]), "v"(w2[0][0]), "v"(w2[0][1]), "v"(w2[0][2]), "v"(w2[1][0]), "v"(w2[1][1]), "v"(w2[1][2]));
#endif
                        __builtin_amdgcn_sched_barrier(0);
                        ldw2(t + 1);                  // (the loads land hundreds of cycles after the last MFMA read these registers)
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                XFH_NOP32_2(acc2[0], acc2[1]);      // idle slots before the VALU code of the stores
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (FUSE == 1) {
                    const int ox = cur.x0 + (l31 & 15);
                    const int oy = cur.y0 + 2 * (j ? br1 : br0) + (l31 >> 4);
                    const int voff = oy < a.H && ox < a.W ? (int)(((size_t)(4 * half) * HW + (size_t)oy * a.W + ox) * 4) : (int)0x80000000;
#pragma unroll
                    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float y = FX ? acc2[m2][r] * FX_SCALE_INV : acc2[m2][r];
                            if (a.relu2) y = fmaxf(y, 0.f);
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rs_out, voff, (int)((m2 * 32 + (r & 3) + 8 * (r >> 2)) * HW * 4), 0);
                        }
                } else {
                    // channels-last: lane (cout l31, half) holds pixels (r & 3) + 8 (r >> 2) + 4 half of the block: 32 lanes = 128 contiguous bytes
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int pm = (r & 3) + 8 * (r >> 2) + 4 * half;          // pixel of the block: row pm >> 4, column pm & 15
                        const int oy = cur.y0 + 2 * (j ? br1 : br0) + (pm >> 4), ox = cur.x0 + (pm & 15);
                        const int voff = oy < a.H && ox < a.W ? (int)((((size_t)oy * a.W + ox) * 64 + l31) * 4) : (int)0x80000000;
#pragma unroll
                        for (int m2 = 0; m2 < 2; ++m2) {
                            float y = FX ? acc2[m2][r] * FX_SCALE_INV : acc2[m2][r];
                            if (a.relu2) y = fmaxf(y, 0.f);
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rs_out, voff, m2 * 128, 0);
                        }
                    }
                }
            }
        }
    };

    Tile cur, nxt;
    int u = u0;
    u += tile_at(u, cur);
    nxt = cur;
    issue_row(0);
    if constexpr (IN_SP) issue_chunk(cur, 0, 0); else issue_loads(cur, 0);
    for (;;) {
        const bool has_next = u < u1;
        if (has_next) u += tile_at(u, nxt);
        BX_STAMP(0)
        if (cur.full) do_tile(std::integral_constant<int, 2>{}, cur, nxt, has_next);
        else do_tile(std::integral_constant<int, 1>{}, cur, nxt, has_next);
        BX_STAMP(50)
        if (!has_next) break;
        dma_barrier();                         // every wave is done with the tile's last tap row before the next chunk is staged
        BX_STAMP(51)
        ++tix;
        cur = nxt;
    }
    XFH_WAIT_VMCNT0();      // the cyclic stream's last DMA must not outlive the workgroup's LDS
#undef BX_STAMP
}


}  // namespace xfh
