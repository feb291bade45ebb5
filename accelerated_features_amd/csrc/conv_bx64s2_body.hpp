// conv_bx64s2x_kernel (k_conv_bx64s2x.hip): the kernel body in a header of its own, so that tests/emu/ can compile the SAME source for the host (XFH_HOST_EMU).
//
// 3x3 stride-2 convolution, 64 -> 64 or 64 -> 128 channels (block4.0 / block5.0; modules/model.py:68,75), on the fp16 matrix cores in the fp16-pair arithmetic
// (bx_split.hpp: two input fragments per pixel, three v_mfma_f32_32x32x16_f16 per K step).  The maps are small (30x40 / 15x20 outputs per VGA frame), so the kernel is
// shaped by balance, not by reuse:
//   * unit = (cout half, image, 16-column strip, 8-row tile): 8x16 output pixels x 64 couts.  VGA batch 64: 768 units for block4.0
//     (three per CU), 512 for block5.0 (two per CU: the second cout half of an image is another unit, not another accumulator);
//   * ONE workgroup of 8 waves per CU (all of its LDS): wave (pb, cb) owns pixel block pb (2 output rows x 16 columns) and cout block
//     cb: one 32x32 accumulator pair, per K step 2 + 3 ds_read_b128 for 3 MFMAs;
//   * the 17x33 input halo of a tile goes through LDS in chunks of 16 channels with EVEN and ODD columns apart
//     ([17 rows, 2688 B apart][parity][17 / 16 pixels, 80 B apart][high parts, low parts][16 channels] fp16): the 16 lanes of a
//     ds_read_b128 group step by two input pixels and would collide pairwise in one plane; 80 B keeps them on distinct banks, and a row pitch that is a multiple of 128 B puts the
//     eight lanes of the block's second output row between the banks of the first row's eight;
//   * TWO such buffers: chunk g + 1 is split and written while chunk g is multiplied.  With all eight waves of a CU in one workgroup
//     nobody else covers a staging phase, and a wave's vector work only hides in the issue gaps of its OWN MFMAs: the five waves that hold staging items
//     split half an item (two pixels x 8 channels) inside each of a chunk's first two tap rows, in the same basic block as that row's 9
//     MFMAs (no branch: lanes without an item write to a dump slot; waves 5 - 7 run a copy of the unit's code without loads and splits).  Raw fp32 values are loaded two chunks ahead (two register sets), also across units;
//   * the split weights (216 KiB per cout half) stream through a THREE-slot LDS ring by LDS-DMA, one slot = one tap row of one chunk
//     (3 K steps, 18 KiB); the DMA of row r + 2 is issued behind the barrier that opens row r -- two rows ahead: a row of 9 MFMAs per wave is
//     ~0.6-1.1 k cycles, the DMA's L2 round trip ~1.6 k (with two slots and one row of look-ahead every row waited ~1 k cycles for its weights:
//     matrix pipe 18-22 % busy).  The DMA is issued by the three waves that hold no staging item (5 - 7), six pieces each, and ONLY they wait for it
//     (vmcnt(6): the row just requested stays in flight); the staging waves' vector-memory queue holds nothing but their own raw loads, which
//     hipcc counts exactly -- they stay in flight across the barriers until the split needs them (with vmcnt(0) at every barrier each chunk's
//     raw loads cost a full memory latency too).  One barrier per tap row, none per chunk.
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
#ifndef XFH_NO_DMA_WAVE
/* marker for tools/check_dma_barriers.py: the barrier that follows is reached by a wave that has issued no LDS-DMA since its last vmcnt(0) (its queue holds compiler-counted loads only) */
#define XFH_NO_DMA_WAVE() asm volatile("; xfh-no-dma-wave" ::: "memory")
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
constexpr int DMA_WAVE0 = 5, NDMA_WAVES = 3;            // the waves without a staging item issue the ring's DMA: NPIECE / 3 pieces each per row
constexpr int LDS_BYTES = 2 * IH * XROWB + NSLOT * SLOT_BYTES + 128 * 4 + 256;      // two X buffers, the weight ring, bias, dump slot
constexpr int NQ = 9;                                   // 4-pixel quads [2 ox0 - 4, 2 ox0 + 32) per halo row
constexpr int NITEM = IH * NQ * 2;                      // (row, quad, 8-channel group)
static_assert(NITEM <= 512 && (2 * IH * XROWB) % 64 == 0 && LDS_BYTES <= 160 * 1024, "one staging item per thread; all of a CU's LDS");
static_assert(NITEM <= DMA_WAVE0 * 64 && NPIECE % NDMA_WAVES == 0, "the DMA waves hold no staging item; equal piece counts (the partial vmcnt is an immediate)");
static_assert(XROWB >= (17 + 16) * PIXB && XROWB % 128 == 0, "row pitch");
}

typedef int i32x4 __attribute__((ext_vector_type(4)));

// NCO: cout halves, 1 (64 couts) or 2 (128).  W4: W % 4 == 0 (no quad straddles the right border: no masking of its tail)
template <int NCO, bool W4>
__device__ __forceinline__ void conv_bx64s2x_body(const Bx64S2xArgs& a) {
    using namespace bx64s2x;
    constexpr int PARB = NEVEN * PIXB, X_BYTES = IH * XROWB, RING_OFF = 2 * X_BYTES, BIAS_OFF = RING_OFF + NSLOT * SLOT_BYTES, DUMP_OFF = BIAS_OFF + 128 * 4;
    constexpr int NXF = 2;                           // input fragments per pixel (high parts, low parts)
    constexpr int CIN = 64, NCH = CIN / 16, NROW = NCH * 3, COUT = 64 * NCO;
    static_assert(NROW % NSLOT == 0 && NCH % 2 == 0, "ring slot, X buffer and register set of a chunk must not depend on the unit");
    XFH_DYN_LDS_BYTES(smem_s2);
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pb = wave >> 1, cb = wave & 1;
    const size_t HW = (size_t)a.H * a.W, HWo = (size_t)a.Ho * a.Wo;
    float* bias_lds = reinterpret_cast<float*>(smem_s2 + BIAS_OFF);
    if (tid < COUT) bias_lds[tid] = a.bias[tid];

    // ---- this workgroup's units: (cout half, image of the list, strip, tile row), rows fastest.  With a batch that is a multiple of 8
    // the images of XCD x are x, x + 8, ... (workgroup id & 7 = XCD): the halo rows shared by neighbouring tiles and both cout halves of
    // an image stay in one L2.
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
    const i32x4 rs_w = make_rsrc(a.wq, (unsigned)(NCO * NROW * SLOT_BYTES));
    const int dma_voff = lane * 16;
    auto lds_addr = [&](const unsigned char* p) { return XFH_LDS_ADDR(p, smem_s2); };
    auto issue_row = [&](int r, int hf) __attribute__((always_inline)) {      // weights of row r (chunk r / 3, tap row r % 3) of cout half hf -> slot r % 3  (waves 5 - 7 only)
#pragma unroll
        for (int k = 0; k < NPIECE / NDMA_WAVES; ++k) {      // (a fixed trip count: no branch between the pieces)
            const int j = wave - DMA_WAVE0 + k * NDMA_WAVES;
            const unsigned m0v = lds_addr(smem_s2 + RING_OFF + (r % NSLOT) * SLOT_BYTES + j * 1024);
            const int soff = (hf * NROW + r) * SLOT_BYTES + j * 1024;
            XFH_DMA_B128_TO_LDS(m0v, dma_voff, rs_w, soff);
        }
    };

    // ---- raw fp32 values of one 16-channel chunk of a tile: item of a thread = 4 consecutive pixels x 8 channels (eight dwordx4 loads,
    // one per channel plane).  Halo column c = 0 .. 32 is image column 2 ox0 - 1 + c; the quads start at 2 ox0 - 4.  Two register sets:
    // chunk g + 2 is loaded (set g & 1) while chunk g + 1 is split (set (g + 1) & 1) and chunk g is multiplied.
    const bool has_item = tid < NITEM;
    const int it_g8 = tid / (IH * NQ), it_row = (tid - it_g8 * (IH * NQ)) / NQ, it_quad = tid % NQ;
    float v[2][8][4];
    int v_gx[2] = {0, 0};                     // first column of the quad (W % 4 != 0: the tail of a quad that straddles the right border is masked)
    // (32-bit offsets, selects instead of branches: the loads sit inside the MFMA block of a row)
    struct LoadAddr { __amdgpu_buffer_rsrc_t rs; int voff; };
    auto load_addr = [&](auto SETC, const Tile& t, bool en) __attribute__((always_inline)) {
        constexpr int S = decltype(SETC)::value;
        LoadAddr la;
        la.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)t.b * CIN * HW), 0, (int)(CIN * HW * sizeof(float)), 0x00020000);
        const int gy = 2 * t.oy0 - 1 + it_row, gx = 2 * t.ox0 - 4 + 4 * it_quad;
        const bool ok = en && has_item && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        v_gx[S] = gx;
        const int off = (it_g8 * 8 * (int)HW + gy * a.W + gx) * 4;
        la.voff = ok ? off : (int)0x80000000;      // (out of range: zeros)
        return la;
    };
    auto load_plane = [&](auto SETC, auto KC, const LoadAddr& la, int chunk) __attribute__((always_inline)) {
        constexpr int S = decltype(SETC)::value, k = decltype(KC)::value;
        const auto q = __builtin_amdgcn_raw_buffer_load_b128(la.rs, la.voff, (chunk * 16 + k) * (int)HW * 4, 0);
        v[S][k][0] = __uint_as_float(q[0]); v[S][k][1] = __uint_as_float(q[1]); v[S][k][2] = __uint_as_float(q[2]); v[S][k][3] = __uint_as_float(q[3]);
    };
    // Half an item (pixels 2 PP, 2 PP + 1 of the quad x 8 channels) of a register set -> an X buffer, as micro-steps that a tap row places
    // behind its MFMAs (S2_FX, S2_P below): split2_f16 of a channel pair of one pixel, ds_write_b128 of a pixel's rows.  Branch-free: a lane without a pixel to
    // write (no item, left of the halo, nothing to stage) writes to the dump slot.
    const int row_base = it_row * XROWB + it_g8 * 16 + 2 * it_quad * PIXB;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 qH[2], qM[2];                                // [pixel of the half] rows of high / low parts: word j = channels 2 j, 2 j + 1

    long long* tr = a.trace && tid == 0 ? a.trace + (size_t)blockIdx.x * 64 : nullptr;
    int tix = 0;
#define S2_STAMP(k) { if (tr && tix == 1) tr[k] = __builtin_amdgcn_s_memtime(); }      /* [0] unit start; row r: [1+4r] start, [2+4r] barrier passed, [3+4r] MFMAs issued; [50] stores issued */
    typedef f16x8 frag_t;
    struct Frag { frag_t x[2]; frag_t w[3]; };      // x[0] = high parts, x[1] = low parts
    unsigned amax = 0;                                // range guard on the converted high parts (bx_split.hpp)
    // lane (pixel l31 of block pb): output row 2 pb + (l31 >> 4), column l31 & 15 -> input row 2 * that (+ dy), even column index = column (+ dx >> 1)
    const int lane_px = 2 * (2 * pb + (l31 >> 4)) * XROWB + (l31 & 15) * PIXB + half * 16;
    f32x16 acc, acc2;

    // ---- one tap row (chunk C, tap row DY) of a unit: barrier, DMA of the next row, (DY = 0) loads of chunk C + 2, then ONE basic block of
    // 9 MFMAs with the fragment reads of the later steps and (DY < 2) the split of half an item of chunk C + 1 in their issue gaps.
    // Every barrier waits for EVERYTHING the wave has in flight (vmcnt(0)).  A first version left "the n youngest" operations in flight -- the eight raw loads
    // behind a row's DMA, the sixteen output stores of the previous unit behind the next unit's first DMA -- on the argument that vmcnt counts in issue order.
    // It does so for loads only: stores are acknowledged out of order with respect to loads, vmcnt(16) was satisfied by early store acks while the DMA was
    // still out, and one unit in ~100 000 read stale weights -- caught by the two-lane soak (tools/lanes_soak.py: 1 wrong image in 1500 concurrent steps), never
    // by a single-stream test.  The loads-only form (vmcnt(8)) was worth 2 us per step and was dropped with it.
    auto row = [&](auto CC, auto DYC, auto STGC, const Tile& cur, const Tile& nxt, bool has_next) __attribute__((always_inline)) {
        constexpr int C = decltype(CC)::value, DY = decltype(DYC)::value, r = C * 3 + DY, P = C & 1;
        constexpr int MODE = decltype(STGC)::value;      // 0: this wave only multiplies (waves 5 - 7); 1: it also loads and splits an item (waves 0 - 4), half in each of a chunk's
        constexpr bool STG = MODE != 0;                  // first two rows.  (Wave 4 -- the last 50 items, on wave 0's SIMD -- doing both halves in the third row instead: slower.)
        if constexpr (MODE == 1) S2_STAMP(1 + 4 * r)      // (the stamps are wave 0's: no store in the DMA waves' copy of the rows)
        // Row 0 of a unit: everything (the previous unit's output stores are in flight, and stores are acknowledged out of order with respect to loads: no
        // partial count is sound while one is out).  Rows 1 - 11: no store has been issued since that wait.  A DMA wave leaves its youngest request -- row r + 1,
        // NPIECE / 3 loads, issued behind the previous barrier -- in flight (loads return in order: row r has landed); a staging wave has issued no DMA at all.
        if constexpr (r == 0) XFH_WAIT_VMCNT0();
        else if constexpr (MODE == 0) XFH_WAIT_VMCNT(NPIECE / NDMA_WAVES);
        else XFH_NO_DMA_WAVE();
        __syncthreads();
        if constexpr (MODE == 1) S2_STAMP(2 + 4 * r)
        // (slot (r + 2) % 3 held row r - 1: every wave's reads of it were waited for before its last MFMAs, in front of this barrier.  The stream is cyclic over the units.)
        if constexpr (MODE == 0) issue_row(r + 2 < NROW ? r + 2 : r + 2 - NROW, r + 2 < NROW ? cur.hf : nxt.hf);
        const unsigned char* wslot = smem_s2 + RING_OFF + (r % NSLOT) * SLOT_BYTES + cb * 3 * 1024 + lane * 16;
        const unsigned char* xrow = smem_s2 + P * X_BYTES + lane_px + DY * XROWB;
        Frag f[2];                             // steps 0 and 1; step 2 is read into f[0] behind the last MFMA of step 0 (slot 6)
        auto load = [&](int s, Frag& o) {          // tap column s: parity s & 1, pixel index + (s >> 1)
#pragma unroll
            for (int q = 0; q < NXF; ++q) o.x[q] = *reinterpret_cast<const frag_t*>(xrow + (s & 1) * PARB + (s >> 1) * PIXB + q * SPLB);
#pragma unroll
            for (int q = 0; q < 3; ++q) o.w[q] = *reinterpret_cast<const frag_t*>(wslot + s * STEP_BYTES + q * 1024);
        };
        load(0, f[0]);
        load(1, f[1]);
        // (DY = 0) raw values of chunk C + 2 (of the next unit for C >= 2) -> set P, free since chunk C - 1 staged it
        constexpr bool same2 = C + 2 < NCH;
        constexpr int SS = P ^ 1;              // (DY < 2) chunk C + 1: set P ^ 1 (loaded during chunk C - 1) -> X buffer P ^ 1 (free since chunk C - 1 was multiplied)
        Tile lt;
        lt.b = same2 ? cur.b : nxt.b; lt.oy0 = same2 ? cur.oy0 : nxt.oy0; lt.ox0 = same2 ? cur.ox0 : nxt.ox0; lt.hf = 0;
        LoadAddr la;
        if constexpr (STG && DY == 0) la = load_addr(std::integral_constant<int, P>{}, lt, same2 || has_next);
        const bool en = C + 1 < NCH || has_next;
        if (r == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; }
        }
        __builtin_amdgcn_sched_barrier(0);
        // 9 fenced slots of { 1 MFMA, one unit of the split (pixel e2, channel pair j: high parts, the two residuals, low parts: ~10 vector ops), a pixel's two ds_write_b128
        // or two planes of raw loads }: a wave's vector work hides in the issue gaps of its OWN MFMAs only, and only if no slot holds more of it than an MFMA takes.
        // Products, small terms first: (q2, xh) (q1, xl) (q0, xh); two accumulators take turns: a dependent MFMA stalls at issue until its predecessor has left the pipe,
        // and blocks the ops behind it.
#define S2_FENCE __builtin_amdgcn_sched_barrier(0);
        // a fragment's registers stay occupied until every MFMA of its step has long been issued: they are not handed to the staging's results
        // while an MFMA may still be reading them (tools/check_mfma_war.py)
#define S2_KEEP(F) XFH_S2_KEEP5(F.x[0], F.x[1], F.w[0], F.w[1], F.w[2]);
#define S2_ON(PP) if constexpr (MODE == 1 && DY == (PP))
        // pixel e = 2 PP + e2 of the quad: halo column c = 4 quad + e - 3 (c < 0: left of the halo), parity c & 1, index (c >> 1) - 2 quad
#define S2_P(PP, E2, Q) S2_ON(PP) { constexpr int e_ = 2 * (PP) + (E2), par_ = (e_ + 1) & 1, idx_ = e_ == 0 ? -2 : e_ == 3 ? 0 : -1; \
        const bool wr_ = en && has_item && !(it_quad == 0 && e_ < 3); \
        *reinterpret_cast<u32x4*>(smem_s2 + (wr_ ? SS * X_BYTES + row_base + par_ * PARB + idx_ * PIXB + (Q) * SPLB : DUMP_OFF)) = (Q) == 0 ? qH[E2] : qM[E2]; }
#define S2_LD(k) if constexpr (STG && DY == 0) load_plane(std::integral_constant<int, P>{}, std::integral_constant<int, k>{}, la, same2 ? C + 2 : C + 2 - NCH);
        {
            constexpr int PP1 = DY & 1;
#define S2_MFX(I) { if constexpr ((I) == 3) { S2_KEEP(f[0]) load(2, f[0]); } constexpr int s_ = ((I) / 3) & 1, j_ = (I) % 3; \
        if constexpr ((I) & 1) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[s_].w[2 - j_], f[s_].x[j_ == 1 ? 1 : 0], acc2, 0, 0, 0); \
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[s_].w[2 - j_], f[s_].x[j_ == 1 ? 1 : 0], acc, 0, 0, 0); }
#define S2_FX(PP, U) S2_ON(PP) { constexpr int e2_ = (U) >> 2, j_ = (U) & 3; float xa_ = v[SS][2 * j_][2 * (PP) + e2_], xb_ = v[SS][2 * j_ + 1][2 * (PP) + e2_]; \
        if (!W4) { const bool z_ = v_gx[SS] + 2 * (PP) + e2_ >= a.W; xa_ = z_ ? 0.f : xa_; xb_ = z_ ? 0.f : xb_; } \
        unsigned hh_, ll_; split2_f16(xa_, xb_, hh_, ll_); fx_track_h(amax, hh_, true); qH[e2_][j_] = hh_; qM[e2_][j_] = ll_; }
            S2_MFX(0) S2_FX(PP1, 0) S2_FENCE
            S2_MFX(1) S2_FX(PP1, 1) S2_FENCE
            S2_MFX(2) S2_FX(PP1, 2) S2_FENCE
            S2_MFX(3) S2_FX(PP1, 3) S2_P(PP1, 0, 0) S2_P(PP1, 0, 1) S2_FENCE
            S2_MFX(4) S2_FX(PP1, 4) S2_LD(0) S2_LD(1) S2_FENCE
            S2_MFX(5) S2_FX(PP1, 5) S2_LD(2) S2_LD(3) S2_FENCE
            S2_MFX(6) S2_FX(PP1, 6) S2_LD(4) S2_LD(5) S2_FENCE
            S2_MFX(7) S2_FX(PP1, 7) S2_LD(6) S2_LD(7) S2_FENCE
            S2_MFX(8) S2_P(PP1, 1, 0) S2_P(PP1, 1, 1) S2_FENCE
#undef S2_MFX
#undef S2_FX
        }
#undef S2_FENCE
#undef S2_ON
#undef S2_P
#undef S2_LD
        S2_KEEP(f[0]) S2_KEEP(f[1])
#undef S2_KEEP
        __builtin_amdgcn_sched_barrier(0);
        XFH_NOP16();      // idle slots: whatever follows must not land in operand registers of the last MFMAs
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE == 1) S2_STAMP(3 + 4 * r)
    };

    auto do_unit = [&](const Tile& cur, const Tile& nxt, bool has_next) __attribute__((always_inline)) {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        auto rows = [&](auto STGC) __attribute__((always_inline)) {
            row(I0{}, I0{}, STGC, cur, nxt, has_next); row(I0{}, I1{}, STGC, cur, nxt, has_next); row(I0{}, I2{}, STGC, cur, nxt, has_next);
            row(I1{}, I0{}, STGC, cur, nxt, has_next); row(I1{}, I1{}, STGC, cur, nxt, has_next); row(I1{}, I2{}, STGC, cur, nxt, has_next);
            row(I2{}, I0{}, STGC, cur, nxt, has_next); row(I2{}, I1{}, STGC, cur, nxt, has_next); row(I2{}, I2{}, STGC, cur, nxt, has_next);
            row(I3{}, I0{}, STGC, cur, nxt, has_next); row(I3{}, I1{}, STGC, cur, nxt, has_next); row(I3{}, I2{}, STGC, cur, nxt, has_next);
        };
        if (wave < DMA_WAVE0) rows(std::integral_constant<int, 1>{});          // (wave-uniform: two copies of the unit's code, no exec masking)
        else rows(std::integral_constant<int, 0>{});
        // ---- bias, ReLU, buffer stores: lane (pixel, half) holds couts 64 hf + 32 cb + (r & 3) + 8 (r >> 2) + 4 half --------------------
        float bs[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 t = *reinterpret_cast<const float4*>(bias_lds + cur.hf * 64 + cb * 32 + 8 * g4 + 4 * half);
            bs[4 * g4] = t.x; bs[4 * g4 + 1] = t.y; bs[4 * g4 + 2] = t.z; bs[4 * g4 + 3] = t.w;
        }
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + ((size_t)cur.b * COUT + cur.hf * 64 + cb * 32) * HWo), 0,
                                                                                (int)(32 * HWo * sizeof(float)), 0x00020000);
        const int oy = cur.oy0 + 2 * pb + (l31 >> 4), ox = cur.ox0 + (l31 & 15);
        const int voff = oy < a.Ho && ox < a.Wo ? (int)(((size_t)(4 * half) * HWo + (size_t)oy * a.Wo + ox) * 4) : (int)0x80000000;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float y = (acc[r] + acc2[r]) * FX_SCALE_INV + bs[r];
            if (a.relu) y = fmaxf(y, 0.f);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rs_out, voff, (int)(((r & 3) + 8 * (r >> 2)) * HWo * 4), 0);
        }
    };

    Tile cur, nxt;
    int u = u0;
    tile_at(u++, cur);
    nxt = cur;
    // prologue: chunk 0 of the first unit is staged with every pipe idle (once per workgroup); chunk 1 waits in set 1
    if (wave >= DMA_WAVE0) { issue_row(0, cur.hf); issue_row(1, cur.hf); }
    {
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        const LoadAddr l0 = load_addr(I0{}, cur, true);
        load_plane(I0{}, std::integral_constant<int, 0>{}, l0, 0); load_plane(I0{}, std::integral_constant<int, 1>{}, l0, 0);
        load_plane(I0{}, std::integral_constant<int, 2>{}, l0, 0); load_plane(I0{}, std::integral_constant<int, 3>{}, l0, 0);
        load_plane(I0{}, std::integral_constant<int, 4>{}, l0, 0); load_plane(I0{}, std::integral_constant<int, 5>{}, l0, 0);
        load_plane(I0{}, std::integral_constant<int, 6>{}, l0, 0); load_plane(I0{}, std::integral_constant<int, 7>{}, l0, 0);
        const LoadAddr l1 = load_addr(I1{}, cur, true);
        load_plane(I1{}, std::integral_constant<int, 0>{}, l1, 1); load_plane(I1{}, std::integral_constant<int, 1>{}, l1, 1);
        load_plane(I1{}, std::integral_constant<int, 2>{}, l1, 1); load_plane(I1{}, std::integral_constant<int, 3>{}, l1, 1);
        load_plane(I1{}, std::integral_constant<int, 4>{}, l1, 1); load_plane(I1{}, std::integral_constant<int, 5>{}, l1, 1);
        load_plane(I1{}, std::integral_constant<int, 6>{}, l1, 1); load_plane(I1{}, std::integral_constant<int, 7>{}, l1, 1);
        // chunk 0 of the first unit: split and written with every pipe idle (once per workgroup)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            u32x4 h, m;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float x0 = v[0][2 * j][e], x1 = v[0][2 * j + 1][e];
                if (!W4) { const bool z = v_gx[0] + e >= a.W; x0 = z ? 0.f : x0; x1 = z ? 0.f : x1; }
                unsigned hh, mm;
                split2_f16(x0, x1, hh, mm);
                fx_track_h(amax, hh, true);
                h[j] = hh; m[j] = mm;
            }
            const bool wr = has_item && !(it_quad == 0 && e < 3);
            const int par = (e + 1) & 1, idx = e == 0 ? -2 : e == 3 ? 0 : -1;
            unsigned char* p = smem_s2 + (wr ? row_base + par * PARB + idx * PIXB : DUMP_OFF);
            *reinterpret_cast<u32x4*>(p) = h;
            *reinterpret_cast<u32x4*>(p + SPLB) = m;
        }
    }
    for (;;) {
        const bool has_next = u < u1;
        if (has_next) tile_at(u++, nxt);
        S2_STAMP(0)
        do_unit(cur, nxt, has_next);
        S2_STAMP(50)
        if (!has_next) break;
        ++tix;
        cur = nxt;
    }
    XFH_WAIT_VMCNT0();      // the cyclic stream's last DMA must not outlive the workgroup's LDS
    fx_report_h(amax, a.status);
#undef S2_STAMP
}


}  // namespace xfh
