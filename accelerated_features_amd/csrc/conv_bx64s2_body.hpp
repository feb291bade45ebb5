// conv_bx64s2x_kernel (k_conv_bx64s2x.hip): the kernel body in a header of its own, so that tests/emu/ can compile the SAME source for the host (XFH_HOST_EMU).
//
// 3x3 stride-2 convolution, 64 -> 64 or 64 -> 128 channels (block4.0 / block5.0; modules/model.py:68,75), on the fp16 matrix cores in the fp16-pair arithmetic
// (bx_split.hpp: two input fragments per pixel, three v_mfma_f32_32x32x16_f16 per K step).  The maps are small (30x40 / 15x20 outputs per VGA frame), so the kernel is
// shaped by balance, not by reuse:
//   * unit = (cout half, image, 16-column strip, 8-row tile): 8x16 output pixels x 64 couts.  VGA batch 64: 768 units for block4.0
//     (three per CU), 512 for block5.0 (two per CU: the second cout half of an image is another unit, not another accumulator);
//   * ONE workgroup per CU (all of its LDS) of TWELVE waves, three per SIMD, SPECIALISED: waves 0 - 7 only multiply -- wave (pb, cb) owns pixel block pb (2 output rows x
//     16 columns) and cout block cb: one 32x32 accumulator pair, per K step 2 + 3 ds_read_b128 for 3 MFMAs -- and waves 8 - 11 (one per SIMD) only stage;
//   * the 17x33 input halo of a tile goes through LDS in chunks of 16 channels with EVEN and ODD columns apart
//     ([17 rows, 2688 B apart][parity][17 / 16 pixels, 80 B apart][high parts, low parts][16 channels] fp16): the 16 lanes of a
//     ds_read_b128 group step by two input pixels and would collide pairwise in one plane; 80 B keeps them on distinct banks, and a row pitch that is a multiple of 128 B puts the
//     eight lanes of the block's second output row between the banks of the first row's eight.  TWO such buffers: chunk g + 1 is split and written while chunk g is multiplied;
//   * the split weights (216 KiB per cout half) stream through a THREE-slot LDS ring by LDS-DMA, one slot = one tap row of one chunk (3 K steps, 18 KiB); the DMA of row
//     r + 2 is issued behind the barrier that opens row r -- two rows ahead: a row is ~0.6-1.1 k cycles, the DMA's L2 round trip ~1.6 k (round 5's two slots and one row
//     of look-ahead: every row waited ~1 k cycles for its weights, matrix pipe 18-22 % busy).  One barrier per tap row, none per chunk.
//
// How it got here (round 6; profiles/r06_s2x_three_slot_trace.txt, r06_s2w_trace.txt).  Round 5's form had eight waves that all multiplied, five of which also split half a
// staging item inside each of a chunk's first two tap rows (hand-placed between their MFMAs).  Its stamps: a row whose waves also split takes 1.2-1.4 k counts against 0.64 k
// for a row that only multiplies -- a wave's vector work does not hide under its OWN MFMAs, it adds to them -- and SIMD 0 carried two of the five staging waves.  Taken apart:
//   (1) three ring slots, the DMA issued (and waited for, partially) by the waves without a staging item:     block4.0 54.8 -> 46.0 us, block5.0 32.1 -> 28.7
//   (2) four multiplying waves (one per SIMD, both cout blocks each) + four staging waves:                      45.5 / 29.2 -- a lone wave does not put its own LDS reads
//       under its own MFMAs either (its row: 18 MFMAs + 24 reads = 1.05 k counts with nothing staged); two multiplying waves per SIMD fill each other's read phases
//   (3) eight multiplying + four staging waves, the staging waves the workgroup's YOUNGEST (this file):        43.7 / 27.0 -- the multiplying waves now wait 0.5-1 k counts
//       per row at the barrier for the staging waves, whose ~60 vector instructions per row crawl beside two older waves that queue on the matrix pipe
//   (4) the same with the staging waves OLDEST:                                                                  45.7 / 29.5 -- they finish early, the multiplying waves lose what they gained
//   (5) (3) with fp16(w) derived in registers instead of read (-20 % LDS reads):                                 43.7 / 27.6 -- not LDS-bound
// A unit takes ~21 k counts in every arrangement with twelve barriers: what the SIMD's arbiter gives one wave it takes from the others.
#pragma once
#ifndef XFH_HOST_EMU
#include "kernels.hpp"
#include <type_traits>
#ifndef XFH_DYN_LDS_BYTES
#define XFH_DYN_LDS_BYTES(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif
#ifndef XFH_LDS_ADDR
#define XFH_LDS_ADDR(p, base) ((unsigned)(size_t)(__attribute__((address_space(3))) void*)(p))
#endif
#ifndef XFH_DMA_B128_TO_LDS
/* LDS-DMA of 16 bytes per lane: M0 = LDS address of the 1-KiB piece, the lane's part of the global address in voff (inline asm: hipcc would make every LDS read wait for all DMA it can see) */
#define XFH_DMA_B128_TO_LDS(m0v, voff, rsrc, soff) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane((int)(m0v))), "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane((int)(soff))) : "memory")      /* (readfirstlane: both are wave-uniform by construction; where hipcc cannot see it, it hands the asm a vector register) */
#endif
#ifndef XFH_WAIT_VMCNT0
#define XFH_WAIT_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
#ifndef XFH_WAIT_VMCNT
#define XFH_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#endif
#ifndef XFH_PIN
#define XFH_PIN(x) asm volatile("" : "+v"(x))
#endif
#ifndef XFH_WAIT_LGKMCNT0
#define XFH_WAIT_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif
#ifndef XFH_DMA_PROTOCOL_EMULATED
/* marker for tools/check_dma_barriers.py: this kernel's waits for its LDS-DMA are partial counts by design; what they guarantee is checked by running the same source with
   the DMA delivered as late as those waits allow (tests/emu/emu.hpp EMU_DEFER_DMA, tests/test_conv_bx64s2_emulated.py), not by the structural lint */
#define XFH_DMA_PROTOCOL_EMULATED() asm volatile("; xfh-dma-protocol-emulated" ::: "memory")
#endif
#define XFH_NOP16() asm volatile("s_nop 7\n\ts_nop 7")
/* five just-read fragments stay occupied up to here (the staging's results are not handed their registers while an MFMA may still be reading them) */
#define XFH_S2_KEEP5(a, b, c, d, e) asm volatile("" :: "v"(a), "v"(b), "v"(c), "v"(d), "v"(e))
#endif
#include "bx_split.hpp"

namespace xfh {

struct Bx64S2xArgs {
    const float* in;
    const void* wq;            // [cout half][cin/16][3 dy][3 dx][2 cout blocks][3 fragments][64 lanes][8 fp16]   (weight_split.hpp: pack_bx64)
    const float* bias;
    float* out;
    int relu, H, W, Ho, Wo, B;
    int nrows, upi;            // 8-row tiles per strip, units per image and cout half
    long long* trace;          // debug: s_memtime stamps of the workgroup's second unit (NULL in production)
    int cold;
    int* status;               // fp16 pair: range guard (bx_split.hpp), may be NULL
};

namespace bx64s2x {
constexpr int SPLB = 32, IH = 17, NEVEN = 17;
constexpr int PIXB = 80;          // bytes per staged pixel: 16 channels x 2 fp16 fragments + 16
constexpr int XROWB = 2688;       // >= (17 + 16) pixels; odd columns of a row behind its even ones
constexpr int STEP_BYTES = 2 * 3 * 1024, SLOT_BYTES = 3 * STEP_BYTES, NPIECE = SLOT_BYTES / 1024;
constexpr int NSLOT = 3;                                // weight ring: row r in slot r % 3, requested two rows ahead
constexpr int LDS_W_BYTES = 2 * IH * XROWB + NSLOT * SLOT_BYTES + 128 * 4 + 1024;   // two X buffers, the weight ring, bias, a KiB for the dummy DMA pieces
constexpr int NQ = 9;                                   // 4-pixel quads [2 ox0 - 4, 2 ox0 + 32) per halo row
constexpr int NITEM = IH * NQ * 2;                      // (row, quad, 8-channel group)
static_assert(NITEM <= 512 && (2 * IH * XROWB) % 64 == 0 && LDS_W_BYTES <= 160 * 1024, "all of a CU's LDS");
static_assert(XROWB >= (17 + 16) * PIXB && XROWB % 128 == 0, "row pitch");
}

typedef int i32x4 __attribute__((ext_vector_type(4)));

// NCO: cout halves, 1 (64 couts) or 2 (128).  W4: W % 4 == 0 (no quad straddles the right border: no masking of its tail)
//   * multiplying wave: its stream is one software pipeline over the unit's 36 steps, the operands of step k + 1 requested in front of the MFMAs of step k; a row's
//     barrier sits between the request of its last step and that step's MFMAs (the reads have completed -- lgkmcnt(0) -- so the ring slot may be overwritten, the MFMAs
//     run on registers while the next row's first operands travel).  The waves issue the weight ring's DMA (3 pieces per wave and row, pieces 18 .. 23 dummies into a
//     dump KiB: one immediate for s_waitcnt) and are the only ones to wait for it: vmcnt(3) leaves the row requested last in flight.  Partial counts are sound here
//     although the wave also stores (the unit's outputs): the THREE youngest loads of its queue are always the pieces of the row requested last, loads return in
//     order, so "at most 3 operations outstanding" implies "every older load has landed" whatever the stores do.  (The unsound form of round 3 counted STORES among the
//     youngest -- "leave the 16 output stores in flight" -- and stores are acknowledged out of order with respect to loads: one unit in ~100 000 read stale weights, caught
//     only by the two-lane soak.)  The wait for a unit's row 0 sits in front of the previous unit's stores (behind them it would be a wait for the stores).
//     What the waits guarantee is CHECKED: tests/test_conv_bx64s2_emulated.py runs this source with the DMA delivered as late as the waits allow (emu.hpp EMU_DEFER_DMA).
//   * staging wave: thread st = tid - 512 holds item st (4 consecutive pixels x 8 channels: eight dwordx4 loads, one per channel plane; halo column c = 0 .. 32 is image
//     column 2 ox0 - 1 + c, the quads start at 2 ox0 - 4) of the chunk's 306 and, if st < 200, one PIXEL of the 50 items left over (item 256 + st / 4, pixel st & 3): 20
//     channel-pair splits per thread and chunk on every staging wave.  Chunk K is split and written while chunk K - 1 is multiplied (pixels 0, 1 | pixel 2 + the extra
//     pixel | pixel 3 in its three rows), raw values travel a chunk ahead (two register sets).  No DMA in these waves: hipcc counts their loads exactly, and they stay in
//     flight across the barriers until the split needs them; the waves pass the same twelve barriers per unit.
// 168 registers per wave (three waves per SIMD), 145 KiB of LDS.
#ifndef XFH_S2_WAIT_PIECES
#define XFH_S2_WAIT_PIECES NPW      /* (tests/test_conv_bx64s2_emulated.py builds the emulation once with 2 * NPW: a count that lets TWO rows stay out must fail under late delivery) */
#endif
template <int NCO, bool W4>
__device__ __forceinline__ void conv_bx64s2w_body(const Bx64S2xArgs& a) {
    using namespace bx64s2x;
    constexpr int PARB = NEVEN * PIXB, X_BYTES = IH * XROWB, RING_OFF = 2 * X_BYTES, BIAS_OFF = RING_OFF + NSLOT * SLOT_BYTES, DUMP_OFF = BIAS_OFF + 128 * 4;
    constexpr int CIN = 64, NCH = CIN / 16, NROW = NCH * 3, COUT = 64 * NCO;
    constexpr int NMW = 8, NPW = 3;                  // multiplying waves (the staging waves come LAST: as the workgroup's oldest waves they slowed the multiplying ones by what they gained); DMA pieces per multiplying wave and row (8 x 3 = 24 >= NPIECE: the last six are dummies)
    static_assert(NROW % NSLOT == 0 && NCH % 2 == 0 && NMW * NPW >= NPIECE && NMW * (NPW - 1) < NPIECE, "ring slot and X buffer of a chunk must not depend on the unit");
    static_assert(DUMP_OFF + 1024 <= LDS_W_BYTES, "the dummy pieces' KiB");
    XFH_DYN_LDS_BYTES(smem_s2);
    XFH_DMA_PROTOCOL_EMULATED();
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t HW = (size_t)a.H * a.W, HWo = (size_t)a.Ho * a.Wo;
    float* bias_lds = reinterpret_cast<float*>(smem_s2 + BIAS_OFF);
    if (tid < COUT) bias_lds[tid] = a.bias[tid];

    int u0, u1, img0, img_step, per_half;
    {
        const int G = (int)gridDim.x, g = (int)blockIdx.x;
        if (xcd_swizzled(a.B) && (G & 7) == 0) {
            per_half = (a.B >> 3) * a.upi;
            const long long U = (long long)NCO * per_half;
            const int slot = g >> 3, nslot = G >> 3;
            u0 = (int)(U * slot / nslot); u1 = (int)(U * (slot + 1) / nslot);
            img0 = g & 7; img_step = 8;
        } else {
            per_half = a.B * a.upi;
            const long long U = (long long)NCO * per_half;
            u0 = (int)(U * g / G); u1 = (int)(U * (g + 1) / G);
            img0 = 0; img_step = 1;
        }
    }
    if (u0 >= u1) return;
    struct Tile { int b, oy0, ox0, hf; };
    auto tile_at = [&](int u, Tile& t) {
        const int hf = u / per_half, rem = u - hf * per_half;
        const int im = rem / a.upi, rem2 = rem - im * a.upi;
        const int col = rem2 / a.nrows, row = rem2 - col * a.nrows;
        t.b = img0 + img_step * im; t.oy0 = row * 8; t.ox0 = col * 16; t.hf = hf;
    };
    typedef f16x8 frag_t;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    long long* tr = a.trace && tid == 0 ? a.trace + (size_t)blockIdx.x * 64 : nullptr;
    int tix = 0;
#define S2W_STAMP(k) { if (tr && tix == 1) tr[k] = __builtin_amdgcn_s_memtime(); }

    if (wave < NMW) {
        // =============================================================== the multiplying waves ===============================================================
        const int mw = wave, pb = mw >> 1, cb = mw & 1;
        auto make_rsrc = [](const void* p, unsigned bytes) {
            const unsigned long long ba = (unsigned long long)p;
            i32x4 r;
            r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ba);
            r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(ba >> 32) & 0xffffu));
            r.z = __builtin_amdgcn_readfirstlane((int)bytes);
            r.w = 0x00020000;
            return r;
        };
        const i32x4 rs_w = make_rsrc(a.wq, (unsigned)(NCO * NROW * SLOT_BYTES));
        auto lds_addr = [&](const unsigned char* p) { return XFH_LDS_ADDR(p, smem_s2); };
        auto issue_row = [&](int r, int hf) __attribute__((always_inline)) {      // weights of row r of cout half hf -> slot r % 3: pieces wave, wave + 8, wave + 16; beyond the slot: zeros into the dump KiB
#pragma unroll
            for (int k = 0; k < NPW; ++k) {
                const int j = mw + NMW * k;
                const bool real = j < NPIECE;
                const unsigned m0v = lds_addr(smem_s2 + (real ? RING_OFF + (r % NSLOT) * SLOT_BYTES + j * 1024 : DUMP_OFF));
                const int soff = real ? (hf * NROW + r) * SLOT_BYTES + j * 1024 : 0;
                const int voff = real ? lane * 16 : (int)0x80000000;
                XFH_DMA_B128_TO_LDS(m0v, voff, rs_w, soff);
            }
        };
        const int lane_px = 2 * (2 * pb + (l31 >> 4)) * XROWB + (l31 & 15) * PIXB + half * 16;
        const unsigned char* xlane = smem_s2 + lane_px;
        const unsigned char* wlane = smem_s2 + RING_OFF + cb * 3 * 1024 + lane * 16;
        struct Frag { frag_t x[2]; frag_t w[3]; };
        auto load = [&](auto RC, auto SC, Frag& o) __attribute__((always_inline)) {      // operands of step (row r, tap column s)
            constexpr int r = decltype(RC)::value, s = decltype(SC)::value, C = r / 3, DY = r % 3;
#pragma unroll
            for (int q = 0; q < 2; ++q) o.x[q] = *reinterpret_cast<const frag_t*>(xlane + (C & 1) * X_BYTES + DY * XROWB + (s & 1) * PARB + (s >> 1) * PIXB + q * SPLB);
#pragma unroll
            for (int q = 0; q < 3; ++q) o.w[q] = *reinterpret_cast<const frag_t*>(wlane + (r % NSLOT) * SLOT_BYTES + s * STEP_BYTES + q * 1024);
        };
        f32x16 acc[2];
        // products, small terms first: (q2, xh) (q1, xl) (q0, xh); two accumulators take turns (a dependent MFMA stalls at issue until its predecessor has left the pipe): MFMA 3 s + j of a row -> acc[(3 s + j) & 1]
        auto mm = [&](auto SC, const Frag& f) __attribute__((always_inline)) {
            constexpr int s = decltype(SC)::value;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[(3 * s + j) & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.w[2 - j], f.x[j == 1 ? 1 : 0], acc[(3 * s + j) & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        Tile cur, nxt;
        int u = u0;
        tile_at(u++, cur);
        nxt = cur;
        issue_row(0, cur.hf);
        issue_row(1, cur.hf);
        XFH_WAIT_VMCNT(NPW);                       // row 0 has landed (this wave's pieces; the others' by the barrier)
        for (;;) {
            const bool has_next = u < u1;
            if (has_next) tile_at(u++, nxt);
            S2W_STAMP(0)
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
            Frag f[2];
            // row r opens: (r > 0) this wave's pieces of row r have landed -- the five loads behind them are row r + 1's --, barrier (everybody's pieces, the chunk's X buffer,
            // and slot (r + 2) % 3 = row r - 1's is free), request row r + 2 (cyclic over the units)
            auto open_row = [&](auto RC) __attribute__((always_inline)) {
                constexpr int r = decltype(RC)::value;
                S2W_STAMP(1 + 4 * r)
                if constexpr (r > 0) XFH_WAIT_VMCNT(XFH_S2_WAIT_PIECES);
                XFH_WAIT_LGKMCNT0();               // (this wave's reads of the slot that is requested next have completed)
                __syncthreads();
                S2W_STAMP(2 + 4 * r)
                issue_row(r + 2 < NROW ? r + 2 : r + 2 - NROW, r + 2 < NROW ? cur.hf : nxt.hf);
            };
            auto row = [&](auto RC) __attribute__((always_inline)) {
                constexpr int r = decltype(RC)::value;
                using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
                using RN = std::integral_constant<int, (r + 1 < NROW ? r + 1 : 0)>;
                // (step k = 3 r + s lives in f[k & 1]: 3 r is odd for odd rows)
                Frag& fa = f[(3 * r) & 1];
                Frag& fb = f[(3 * r + 1) & 1];
                load(RC, I1{}, fb); mm(I0{}, fa);        // step 0 multiplied, step 1 travelling
                load(RC, I2{}, fa); mm(I1{}, fb);        // step 1 multiplied, step 2 travelling
                if constexpr (r + 1 < NROW) { open_row(RN{}); load(RN{}, I0{}, fb); }      // the next row opens between the request of this row's last operands and their MFMAs
                mm(I2{}, fa);
                S2W_STAMP(3 + 4 * r)
            };
            open_row(std::integral_constant<int, 0>{});
            load(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, f[0]);
            row(std::integral_constant<int, 0>{}); row(std::integral_constant<int, 1>{}); row(std::integral_constant<int, 2>{}); row(std::integral_constant<int, 3>{});
            row(std::integral_constant<int, 4>{}); row(std::integral_constant<int, 5>{}); row(std::integral_constant<int, 6>{}); row(std::integral_constant<int, 7>{});
            row(std::integral_constant<int, 8>{}); row(std::integral_constant<int, 9>{}); row(std::integral_constant<int, 10>{}); row(std::integral_constant<int, 11>{});
            XFH_S2_KEEP5(f[0].x[0], f[0].x[1], f[1].x[0], f[1].x[1], f[0].w[0]);
            XFH_NOP16();      // idle slots: whatever follows must not land in operand registers of the last MFMAs
            __builtin_amdgcn_sched_barrier(0);
            // ---- bias, ReLU, buffer stores: lane (pixel, half) holds couts 64 hf + 32 cb + (r & 3) + 8 (r >> 2) + 4 half.  The NEXT unit's row 0 (requested two rows ago) is
            // waited for HERE, in front of the stores: behind them the same count would wait for the stores
            XFH_WAIT_VMCNT(NPW);
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + ((size_t)cur.b * COUT + cur.hf * 64 + cb * 32) * HWo), 0, (int)(32 * HWo * sizeof(float)), 0x00020000);
            const int oy = cur.oy0 + 2 * pb + (l31 >> 4), ox = cur.ox0 + (l31 & 15);
            const int voff = oy < a.Ho && ox < a.Wo ? (int)(((size_t)(4 * half) * HWo + (size_t)oy * a.Wo + ox) * 4) : (int)0x80000000;
            float bs[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 t = *reinterpret_cast<const float4*>(bias_lds + cur.hf * 64 + cb * 32 + 8 * g4 + 4 * half);
                bs[4 * g4] = t.x; bs[4 * g4 + 1] = t.y; bs[4 * g4 + 2] = t.z; bs[4 * g4 + 3] = t.w;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float y = (acc[0][r] + acc[1][r]) * FX_SCALE_INV + bs[r];
                if (a.relu) y = fmaxf(y, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rs_out, voff, (int)(((r & 3) + 8 * (r >> 2)) * HWo * 4), 0);
            }
            S2W_STAMP(50)
            if (!has_next) break;
            ++tix;
            cur = nxt;
        }
        XFH_WAIT_VMCNT0();      // the cyclic stream's last DMA must not outlive the workgroup's LDS
    } else {
        // ================================================================= the staging waves =================================================================
        const int st = tid - 64 * NMW;
        // item st of a chunk (4 consecutive pixels x 8 channels: eight dwordx4 loads, one per channel plane) ...
        const int it_g8 = st / (IH * NQ), it_rem = st - it_g8 * (IH * NQ), it_row = it_rem / NQ, it_quad = it_rem - it_row * NQ;
        // ... and one pixel of the items 256 .. 305 (all of channel group 1): item 256 + st / 4, pixel st & 3
        const bool has_x = st < 4 * (NITEM - 256);
        const int x_item = 256 + (st >> 2), x_e = st & 3, x_rem = x_item - IH * NQ, x_row = x_rem / NQ, x_quad = x_rem - x_row * NQ;
        static_assert(NITEM > 256 && NITEM - 256 <= 64 && 256 >= IH * NQ, "the left-over items all belong to channel group 1");
        float v[2][8][4];                          // raw values of the item, two chunks in flight
        float vx[8];                               // ... and of the extra pixel (one set: requested in the row after the one that split it, two rows ahead of its use)
        int v_gx[2] = {0, 0};
        unsigned amax = 0;                         // range guard on the converted high parts (bx_split.hpp)
        auto load_chunk = [&](auto SETC, const Tile& t, bool en, int chunk) __attribute__((always_inline)) {
            constexpr int S = decltype(SETC)::value;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)t.b * CIN * HW), 0, (int)(CIN * HW * sizeof(float)), 0x00020000);
            const int gy = 2 * t.oy0 - 1 + it_row, gx = 2 * t.ox0 - 4 + 4 * it_quad;
            const bool ok = en && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            v_gx[S] = gx;
            const int voff = ok ? (it_g8 * 8 * (int)HW + gy * a.W + gx) * 4 : (int)0x80000000;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const auto q = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (chunk * 16 + k) * (int)HW * 4, 0);
                v[S][k][0] = __uint_as_float(q[0]); v[S][k][1] = __uint_as_float(q[1]); v[S][k][2] = __uint_as_float(q[2]); v[S][k][3] = __uint_as_float(q[3]);
            }
        };
        auto load_x = [&](const Tile& t, bool en, int chunk) __attribute__((always_inline)) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)t.b * CIN * HW), 0, (int)(CIN * HW * sizeof(float)), 0x00020000);
            const int gyx = 2 * t.oy0 - 1 + x_row, gxx = 2 * t.ox0 - 4 + 4 * x_quad + x_e;
            const bool okx = en && has_x && gyx >= 0 && gyx < a.H && gxx >= 0 && gxx < a.W;      // (a column beyond the right border reads zeros here: no masking of the extra pixel)
            const int voffx = okx ? (8 * (int)HW + gyx * a.W + gxx) * 4 : (int)0x80000000;
#pragma unroll
            for (int k = 0; k < 8; ++k) vx[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voffx, (chunk * 16 + k) * (int)HW * 4, 0));
        };
        // pixel e of quad q: halo column c = 4 q + e - 3 (c < 0: left of the halo), parity c & 1, index c >> 1.  One register per thread (its quad's base) + immediates:
        // the pin keeps hipcc from keeping a register per (pixel, buffer) across the unit (168 registers per wave: three waves per SIMD)
        const int it_base = it_row * XROWB + it_g8 * 16 + 2 * it_quad * PIXB, x_base = x_row * XROWB + 16 + 2 * x_quad * PIXB;
        auto px_off = [](int e) { return ((e + 1) & 1) * PARB + (e == 0 ? -2 : e == 3 ? 0 : -1) * PIXB; };
        auto stage_px = [&](auto SETC, auto EC, int buf, bool en) __attribute__((always_inline)) {      // pixel E of this thread's item: four channel pairs -> a row of high parts, a row of low parts
            constexpr int S = decltype(SETC)::value, e = decltype(EC)::value;
            u32x4 h, m;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x0 = v[S][2 * j][e], x1 = v[S][2 * j + 1][e];
                if (!W4) { const bool z = v_gx[S] + e >= a.W; x0 = z ? 0.f : x0; x1 = z ? 0.f : x1; }
                unsigned hh, mm_;
                split2_f16(x0, x1, hh, mm_);
                fx_track_h(amax, hh, true);
                h[j] = hh; m[j] = mm_;
            }
            if (en && !(it_quad == 0 && e < 3)) {
                int base = it_base;
                XFH_PIN(base);
                unsigned char* p = smem_s2 + base + (buf * X_BYTES + px_off(e));
                *reinterpret_cast<u32x4*>(p) = h;
                *reinterpret_cast<u32x4*>(p + SPLB) = m;
            }
        };
        auto stage_x = [&](int buf, bool en) __attribute__((always_inline)) {
            u32x4 h, m;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned hh, mm_;
                split2_f16(vx[2 * j], vx[2 * j + 1], hh, mm_);
                fx_track_h(amax, hh, true);
                h[j] = hh; m[j] = mm_;
            }
            if (en && has_x && !(x_quad == 0 && x_e < 3)) {
                int base = x_base + px_off(x_e);
                XFH_PIN(base);
                unsigned char* p = smem_s2 + base + buf * X_BYTES;
                *reinterpret_cast<u32x4*>(p) = h;
                *reinterpret_cast<u32x4*>(p + SPLB) = m;
            }
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        Tile cur, nxt;
        int u = u0;
        tile_at(u++, cur);
        nxt = cur;
        // prologue: chunk 0 of the first unit is staged with every pipe idle (once per workgroup); chunk 1 waits in set 1
        load_chunk(I0{}, cur, true, 0);
        load_x(cur, true, 0);
        load_chunk(I1{}, cur, true, 1);
        stage_px(I0{}, I0{}, 0, true); stage_px(I0{}, I1{}, 0, true); stage_px(I0{}, I2{}, 0, true); stage_px(I0{}, I3{}, 0, true);
        stage_x(0, true);
        load_x(cur, true, 1);
        for (;;) {
            const bool has_next = u < u1;
            if (has_next) tile_at(u++, nxt);
            // row (chunk C, tap row DY): barrier; then a third of chunk C + 1's staging (set and X buffer (C + 1) & 1); at DY = 0 first the loads of chunk C + 2 into set C & 1,
            // free since chunk C was staged
            auto srow = [&](auto CC, auto DYC) __attribute__((always_inline)) {
                constexpr int C = decltype(CC)::value, DY = decltype(DYC)::value, SS = (C + 1) & 1;
                __syncthreads();
                const bool en = C + 1 < NCH || has_next;
                if constexpr (DY == 0) {
                    constexpr bool same2 = C + 2 < NCH;
                    load_chunk(std::integral_constant<int, (C & 1)>{}, same2 ? cur : nxt, same2 || has_next, same2 ? C + 2 : C + 2 - NCH);
                    stage_px(std::integral_constant<int, SS>{}, I0{}, SS, en); stage_px(std::integral_constant<int, SS>{}, I1{}, SS, en);
                } else if constexpr (DY == 1) {
                    stage_px(std::integral_constant<int, SS>{}, I2{}, SS, en); stage_x(SS, en);
                } else {
                    constexpr bool same2 = C + 2 < NCH;
                    load_x(same2 ? cur : nxt, same2 || has_next, same2 ? C + 2 : C + 2 - NCH);      // (the extra pixel of chunk C + 2: its registers are free since row (C, 1))
                    stage_px(std::integral_constant<int, SS>{}, I3{}, SS, en);
                }
            };
            srow(I0{}, I0{}); srow(I0{}, I1{}); srow(I0{}, I2{});
            srow(I1{}, I0{}); srow(I1{}, I1{}); srow(I1{}, I2{});
            srow(I2{}, I0{}); srow(I2{}, I1{}); srow(I2{}, I2{});
            srow(I3{}, I0{}); srow(I3{}, I1{}); srow(I3{}, I2{});
            if (!has_next) break;
            cur = nxt;
        }
        fx_report_h(amax, a.status);
    }
#undef S2W_STAMP
}


}  // namespace xfh
