// One linear layer y = act(x W + b) of the fine_matcher (modules/model.py:97-111: 128 -> 512 -> 512 -> 512 -> 512 -> 64 on every match of match_xfeat_star) in the
// fp16-pair arithmetic (bx_split.hpp): three v_mfma_f32_32x32x16_f16 per K = 16 where the f32 matrix cores take eight v_mfma_f32_32x32x2_f32.
//
// A workgroup of 4 waves takes 256 rows x 64 columns (wave w: rows 64 w .. + 63), K in chunks of 32.  Between the layers of the chain the activations travel in the SPLIT
// form: row r = [K halves xh | K halves xl] in the bytes the fp32 row had (the same workspace) -- the producing layer splits each value ONCE in its epilogue, where the
// first form of this kernel (fp32 in, fp32 out) split it in every one of the eight column blocks that read the row, 100 vector instructions per chunk next to 24 MFMAs.
//
// The loop (measured on the first form, FINDINGS "linear_fx"): two workgroups share a CU and run in lock-step (same code, same start), so what one wave does NOT overlap with
// its own MFMAs is not overlapped at all.  Hence: the LDS tile is double-buffered (one barrier per chunk), the next chunk's rows are requested one whole chunk ahead and are
// written to the other buffer BETWEEN this chunk's two MFMA groups, and the weight fragments roll (group 0's registers are refilled for the next chunk as soon as group 0
// has been issued, group 1's behind group 1) -- every wait is for the OLDEST request in flight only, and no request sits behind a branch (there the compiler loses count
// and waits for everything, the prefetch it has just issued included: 47 % of the first form's wave cycles).
//
// MFMA orientation: A = W (lane = column), B = x (lane = row): D holds, per lane, ONE row and 4 x 4 consecutive columns -- the epilogue packs four values into 8 bytes
// of each plane, transposes through the (now idle) LDS and stores full 128-byte lines.
#pragma once
#ifndef XFH_HOST_EMU
#include "common.hpp"
#endif
#include "bx_split.hpp"

#ifndef XFH_DYN_LDS_BYTES
#define XFH_DYN_LDS_BYTES(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif
#ifndef XFH_AGPR
#define XFH_AGPR(x) asm volatile("" : "+a"(x))
#endif
#ifndef XFH_SCHED_FENCE
#define XFH_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#ifndef XFH_LDS_ADDR
#define XFH_LDS_ADDR(p, base) ((unsigned)(size_t)(__attribute__((address_space(3))) void*)(p))
#endif
#ifndef XFH_DMA_B128_TO_LDS
/* LDS-DMA of 16 bytes per lane: M0 = LDS address of the 1-KiB piece, the lane's part of the global address in voff (inline asm: hipcc would make every LDS read wait for all DMA it can see) */
#define XFH_DMA_B128_TO_LDS(m0v, voff, rsrc, soff) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane((int)(m0v))), "v"(voff), "s"(rsrc), "s"(__builtin_amdgcn_readfirstlane((int)(soff))) : "memory")
#endif
#ifndef XFH_WAIT_VMCNT0
#define XFH_WAIT_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
#ifndef XFH_WAIT_VMCNT
#define XFH_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#endif

namespace xfh {

enum LinFxIn { LFX_IN_F32 = 0, LFX_IN_GATHER2 = 2, LFX_IN_PAIR = 3 };      // (0 / 2: LinLoader's LOAD_ROWMAJOR / LOAD_GATHER2)
enum LinFxOut { LFX_OUT_F32 = 0, LFX_OUT_PAIR = 1 };

struct LinFxArgs {
    const uint4* wq;            // weight_split.hpp: pack_linear_fx
    const float* bias;          // padded to n_pad
    int N, relu;
    const void* x;              // F32: (M, ldx) fp32 ; PAIR: rows of [ldx halves xh | ldx halves xl] ; GATHER2: desc0 (P, cap, 64)
    int ldx;
    const float* x2;            // GATHER2: desc1
    const int64_t* idx0;
    const int64_t* idx1;
    const int32_t* rowmap;      // GATHER2: compact row -> p * cap + r
    int cap;
    int M;
    const int32_t* m_dev;       // optional live row count (<= M)
    void* y;                    // F32: (M, ldy) fp32 ; PAIR: rows of [ldy halves | ldy halves]
    int ldy;
    int* status;                // bit 0: a value left the fp16 range (bx_split.hpp)
    int n_row_blocks, n_col_blocks;
};

namespace linfx {
constexpr int ROWS = 256, KC = 32;
constexpr int RB = 144;                     // LDS row: [32 halves xh | 32 halves xl | 16 bytes]: the 16 rows of a ds_read_b128 pass start 36 banks apart -- conflict-free
constexpr int BUF = ROWS * RB;              // 36864
constexpr int LDS_BYTES = 2 * BUF;          // 73728: two workgroups per CU
constexpr int OUT_RB = 272;                 // epilogue row: [64 halves yh | 64 halves yl | 16 bytes]
static_assert(ROWS * OUT_RB <= LDS_BYTES, "the epilogue's transposition reuses the staging buffers");
typedef unsigned u4 __attribute__((ext_vector_type(4)));
}  // namespace linfx

template <int K, int IN, int OUT>
__device__ __forceinline__ void linear_fx_body(const LinFxArgs& a) {
    using namespace linfx;
    constexpr int NKC = K / KC, NS = K / 16;
    static_assert(K % KC == 0 && NKC >= 2, "K: a multiple of 32, at least 64");
    XFH_DYN_LDS_BYTES(smem_lf);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    // XCD-aware mapping (common.hpp): the column blocks of ONE row block run on one XCD, next to each other in time -- its rows come from HBM once, from that XCD's L2 for the others
    int rbk, nb;
    if (!xcd_group_map((int)blockIdx.x, a.n_col_blocks, a.n_row_blocks, rbk, nb)) return;      // (n_row_blocks is padded to a multiple of 8: the padding workgroups leave)
    const int row0 = rbk * ROWS, n0 = nb * 64;
    int Mlive = a.M;
    if (a.m_dev) Mlive = min(a.M, *a.m_dev);
    if (row0 >= Mlive) return;

    // ---- staging: this thread's share of a chunk (256 rows x 32 k).  Rows past the live count read the LAST live row instead (an output row depends on its own input row
    // only and the epilogue stores live rows only): the loads carry no condition
    constexpr int NLD = 8;
    long src0[IN == LFX_IN_PAIR ? 4 : 8], src1[IN == LFX_IN_GATHER2 ? 8 : 1];
    if constexpr (IN == LFX_IN_PAIR) {
#pragma unroll
        for (int i = 0; i < 4; ++i) src0[i] = (long)min(row0 + (tid >> 2) + 64 * i, Mlive - 1) * (2 * a.ldx) + 8 * (tid & 3);      // (halves)
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = min(row0 + (tid >> 3) + 32 * i, Mlive - 1);
            if constexpr (IN == LFX_IN_F32) src0[i] = (long)row * a.ldx + 4 * (tid & 7);
            else {
                const int code = a.rowmap[row];
                const int p = code / a.cap;
                src0[i] = ((long)p * a.cap + (long)a.idx0[code]) * 64 + 4 * (tid & 7);
                src1[i] = ((long)p * a.cap + (long)a.idx1[code]) * 64 + 4 * (tid & 7);
            }
        }
    }
    u4 xr[NLD];
    auto stage_load = [&](int kc) {
        if constexpr (IN == LFX_IN_PAIR) {
            const _Float16* xp = static_cast<const _Float16*>(a.x);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xr[2 * i] = *reinterpret_cast<const u4*>(xp + src0[i] + kc * KC);
                xr[2 * i + 1] = *reinterpret_cast<const u4*>(xp + src0[i] + a.ldx + kc * KC);
            }
        } else {
            const float* xp = static_cast<const float*>(a.x);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (IN == LFX_IN_F32) xr[i] = *reinterpret_cast<const u4*>(xp + src0[i] + kc * KC);
                else {      // the row is [64 of desc0 | 64 of desc1]: one load through a selected address (a load in each arm of a branch: see above)
                    const float* p = kc < 2 ? xp + src0[i] + kc * KC : a.x2 + src1[i] + (kc - 2) * KC;
                    xr[i] = *reinterpret_cast<const u4*>(p);
                }
            }
        }
    };
    unsigned amax = 0;
    auto stage_store = [&](unsigned char* buf) {
        if constexpr (IN == LFX_IN_PAIR) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned char* p = buf + ((tid >> 2) + 64 * i) * RB + 16 * (tid & 3);
                *reinterpret_cast<u4*>(p) = xr[2 * i];
                *reinterpret_cast<u4*>(p + 64) = xr[2 * i + 1];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint2 h, l;
                split2_f16_scalar(__uint_as_float(xr[i][0]), __uint_as_float(xr[i][1]), h.x, l.x);
                split2_f16_scalar(__uint_as_float(xr[i][2]), __uint_as_float(xr[i][3]), h.y, l.y);
                fx_track_h(amax, h.x, true); fx_track_h(amax, h.y, true);
                unsigned char* p = buf + ((tid >> 3) + 32 * i) * RB + 8 * (tid & 7);
                *reinterpret_cast<uint2*>(p) = h;
                *reinterpret_cast<uint2*>(p + 64) = l;
            }
        }
    };
    // ---- weight fragments of one K step: [fragment q0 q1 q2][column half-block], straight from global memory (L2-resident: 1.5 MB per 512 x 512 layer) in operand order
    const uint4* wp = a.wq + (size_t)nb * NS * 6 * 64 + lane;
    f16x8 w0[3][2], w1[3][2];
    auto wload = [&](f16x8 (&w)[3][2], int s) {
#pragma unroll
        for (int sp = 0; sp < 3; ++sp)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) w[sp][cb] = __builtin_bit_cast(f16x8, wp[(((size_t)s * 3 + sp) * 2 + cb) * 64]);
    };
    f32x16 acc[2][2];      // [row half-block][column half-block]
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;
    // q2 xh + q1 xl + q0 xh (bx_split.hpp), the four accumulators in turn.  Every matrix operand lives in an accumulation register (the loads write them there directly):
    // registers no vector-ALU result is ever allocated to, so none can land in an operand the matrix core is still reading (DESIGN 3.1 / tools/check_mfma_war.py)
    // q2 xh + q0 xh + q1 xl (bx_split.hpp) per K step, the four accumulators in turn.  Every matrix operand lives in an accumulation register (the loads write them there
    // directly): registers no vector-ALU result is ever allocated to, so none can land in an operand the matrix core is still reading (DESIGN 3.1 / tools/check_mfma_war.py).
    // Two waves per SIMD leave 128 of them: 64 accumulators, 48 fragments and ONE K step's x operands (16) -- xh is free after a step's first eight MFMAs and xl is wanted
    // for its last four only, so the next step's operands are read behind MFMA 8 (xh) and MFMA 12 (xl) and are there when they are wanted.
    f16x8 ah[2], al[2];      // [row half-block]
    auto mfma4 = [&](f16x8 (&wf)[2], f16x8 (&xo)[2]) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cb], xo[rb], acc[rb][cb], 0, 0, 0);
    };
    auto pin_w = [&](f16x8 (&w)[3][2]) {
#pragma unroll
        for (int sp = 0; sp < 3; ++sp)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) XFH_AGPR(w[sp][cb]);
    };
    // One chunk.  Request order (the waits are for the oldest requests only): [rows of chunk kc + 2] during group 0, [group 0's fragments of chunk kc + 1] in front of
    // group 1, [group 1's fragments] behind it.
    auto chunk = [&](int kc, auto last) {
        const unsigned char* cur = smem_lf + (kc & 1) * BUF + (wave * 64 + l31) * RB + half * 16;
        auto read_h = [&](int st) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) ah[rb] = *reinterpret_cast<const f16x8*>(cur + rb * 32 * RB + st * 32);
        };
        auto read_l = [&](int st) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) al[rb] = *reinterpret_cast<const f16x8*>(cur + rb * 32 * RB + st * 32 + 64);
        };
        read_h(0); read_l(0);
        pin_w(w0);
        XFH_AGPR(ah[0]); XFH_AGPR(ah[1]); XFH_AGPR(al[0]); XFH_AGPR(al[1]);      // (no pin inside a group: an asm statement ends the region the issue order below is laid out in)
        // (the writes stand in front of the second step's reads in program order: the compiler cannot tell the two buffers apart and keeps a write behind every read it follows)
        if constexpr (!decltype(last)::value) stage_store(smem_lf + ((kc + 1) & 1) * BUF);      // (the rows requested one chunk ago)
        mfma4(w0[2], ah); mfma4(w0[0], ah);
        read_h(1);
        mfma4(w0[1], al);
        read_l(1);
        if constexpr (!decltype(last)::value) {
            stage_load(min(kc + 2, NKC - 1));                  // (the last chunk is requested twice: no condition on a load)
#ifndef XFH_HOST_EMU
            // the issue order of group 0: the LDS writes behind its first eight MFMAs, two row requests behind each of the last four
            // (sched_group_barrier masks: 0x8 MFMA, 0x20 VMEM read, 0x100 DS read, 0x200 DS write)
            constexpr int W1 = (IN == LFX_IN_PAIR ? 8 : 16) / 8;
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, W1, 0); }
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x20, 2, 0); }
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#endif
        }
        XFH_SCHED_FENCE();
        if constexpr (!decltype(last)::value) wload(w0, 2 * (kc + 1));
        pin_w(w1);
        XFH_AGPR(ah[0]); XFH_AGPR(ah[1]); XFH_AGPR(al[0]); XFH_AGPR(al[1]);
        mfma4(w1[2], ah); mfma4(w1[0], ah);
        mfma4(w1[1], al);
        if constexpr (!decltype(last)::value) wload(w1, 2 * (kc + 1) + 1);
        else XFH_SCHED_FENCE();      // (the epilogue's arithmetic stays out of the last group: nothing of it may land in an operand register still being read)
        if constexpr (!decltype(last)::value) __syncthreads();      // everybody has read this chunk's buffer (the chunk after the next is written there) and written the next chunk's
    };
    stage_load(0);
    stage_store(smem_lf);
    stage_load(1);
    XFH_SCHED_FENCE();      // (the loop's request order: rows, group 0's fragments, group 1's -- its first wait counts on it)
    wload(w0, 0);
    XFH_SCHED_FENCE();
    wload(w1, 1);
    __syncthreads();
#pragma unroll 1
    for (int kc = 0; kc < NKC - 1; ++kc) chunk(kc, std::false_type{});
    chunk(NKC - 1, std::true_type{});

    // ---- epilogue.  D of A = W, B = x: lane (row l31, half) holds columns (r & 3) + 8 (r >> 2) + 4 half of the 32-column half-block
    if constexpr (OUT == LFX_OUT_PAIR) {
        __syncthreads();      // (the last chunk's reads)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = cb * 32 + 8 * g + 4 * half;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = fmaf(acc[rb][cb][4 * g + j], FX_SCALE_INV, a.bias[n0 + c0 + j]);
                        if (a.relu) v[j] = fmaxf(v[j], 0.f);
                    }
                    uint2 h, l;
                    split2_f16_scalar(v[0], v[1], h.x, l.x);
                    split2_f16_scalar(v[2], v[3], h.y, l.y);
                    fx_track_h(amax, h.x, true); fx_track_h(amax, h.y, true);
                    unsigned char* p = smem_lf + (wave * 64 + rb * 32 + l31) * OUT_RB + 2 * c0;
                    *reinterpret_cast<uint2*>(p) = h;
                    *reinterpret_cast<uint2*>(p + 128) = l;
                }
        __syncthreads();
        _Float16* yp = static_cast<_Float16*>(a.y);
#pragma unroll
        for (int ps = 0; ps < 16; ++ps) {      // 16 lanes per row: eight 16-byte pieces of yh, eight of yl -- two full lines per row
            const int rl = ps * 16 + (tid >> 4), piece = tid & 15;
            const u4 v = *reinterpret_cast<const u4*>(smem_lf + rl * OUT_RB + 16 * piece);
            if (row0 + rl < Mlive) *reinterpret_cast<u4*>(yp + (long)(row0 + rl) * (2 * a.ldy) + (piece >> 3) * a.ldy + n0 + 8 * (piece & 7)) = v;
        }
    } else {
        float* yp = static_cast<float*>(a.y);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const int row = row0 + wave * 64 + rb * 32 + l31;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = n0 + cb * 32 + 8 * g + 4 * half;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = fmaf(acc[rb][cb][4 * g + j], FX_SCALE_INV, a.bias[c0 + j]);
                        if (a.relu) v[j] = fmaxf(v[j], 0.f);
                    }
                    if (row < Mlive && c0 < a.N) *reinterpret_cast<float4*>(yp + (long)row * a.ldy + c0) = make_float4(v[0], v[1], v[2], v[3]);      // (N, ldy: multiples of 4 -- the launcher checks)
                }
        }
    }
    fx_report_h(amax, a.status);
}


// ------------------------------------------------------------------------------------------------------------------------------------------------------------------------
// The chain's inner layers (pair in, pair out: 512 -> 512, three of the five and 85 % of the arithmetic) with EVERY operand brought in by LDS-DMA.  Measured on the form above
// (profiles/r05_pmc_dense_linear_fx2.txt): its waves wait for an instruction to ISSUE 68 % of their cycles with the matrix pipes 29 % busy -- 160 global-load wave-instructions
// per CU and chunk round (the fragments alone 96: each of a workgroup's four waves fetches the same 12 KB), 5300 cycles per round where the MFMAs need 1536.  Here a
// workgroup of EIGHT waves takes 256 rows x 128 columns (wave: row group w & 3, column half w >> 2: two waves per SIMD, one workgroup per CU): per chunk its 32 KB of rows
// and its 24 fragments arrive as 56 one-KiB DMA pieces (7 per wave) -- a third of the requests per MFMA, no staging registers, no LDS stores (13 cycles each: the LDS reads
// that replace the fragments' global loads take 4).
//
// LDS stage (two of them): [x: 256 rows x (4 pieces xh | 4 pieces xl), swizzled][24 fragments of 1 KiB: column half 2, K step 2, fragment 3, column half-block 2].
// The DMA writes piece i of a KiB at 16 i, so the swizzle is in WHICH piece a lane fetches: piece (row, q) lives at 256 (row >> 1) + 16 ((8 (row & 1) + q) ^ ((row >> 1) & 15)) --
// the 16 lanes of a ds_read_b128 pass (MI355X_MICROARCH: {0-3, 12-15, 20-27}, ...) then hit 16 different 16-byte bank groups.  Rows past the live count: outside the
// resource's range, the DMA writes zeros.
//
// Prefetch depth (profiles/r05_pmc_fine_dma.txt, the two-stage form: 225 us per layer at 2.2 TB/s of HBM traffic, waves 40 % at s_waitcnt): the rows of a row block are
// fetched from HBM by whichever of its column blocks asks first and everybody waits for that miss -- one chunk of compute (0.8 us) does not cover it.  The rows are therefore
// requested two chunks ahead (three stages), the fragments (L2-resident) one chunk ahead; the request order [fragments, rows] makes the end-of-chunk wait vmcnt(4): everything
// but the four row pieces just issued.
#ifndef XFH_LFXD_TRACE
#define XFH_LFXD_TRACE 0
#endif
namespace linfxd {
constexpr int ROWS = 256, COLS = 128, KC = 32, THREADS = 512;
constexpr int XB = ROWS * 128;              // 32768: the rows of a chunk
constexpr int NXS = 3;                      // ... in a ring of three: requested TWO chunks ahead (see below)
constexpr int WB = 24 * 1024;               // the fragments of a chunk, two stages
constexpr int WOFF = NXS * XB;
constexpr int LDS_BYTES = WOFF + 2 * WB;    // 147456: one workgroup per CU
static_assert(ROWS * linfx::OUT_RB <= LDS_BYTES, "the epilogue's transposition (one column half at a time) reuses the stages");
typedef int i4 __attribute__((ext_vector_type(4)));
}  // namespace linfxd

template <int K>
__device__ __forceinline__ void linear_fxd_body(const LinFxArgs& a) {
    using namespace linfxd;
    using linfx::OUT_RB;
    using linfx::u4;
    constexpr int NKC = K / KC, NS = K / 16;
    XFH_DYN_LDS_BYTES(smem_ld);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int rg = wave & 3, ch = wave >> 2;
    int rbk, nbk;
    if (!xcd_group_map((int)blockIdx.x, a.n_col_blocks, a.n_row_blocks, rbk, nbk)) return;      // (n_col_blocks: blocks of 128 columns)
    const int row0 = rbk * ROWS, n0 = nbk * COLS;
    int Mlive = a.M;
    if (a.m_dev) Mlive = min(a.M, *a.m_dev);
    if (row0 >= Mlive) return;

    auto make_rsrc = [](const void* p, unsigned bytes) {
        const unsigned long long ba = (unsigned long long)p;
        i4 r;
        r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ba);
        r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(ba >> 32) & 0xffffu));
        r.z = __builtin_amdgcn_readfirstlane((int)bytes);
        r.w = 0x00020000;
        return r;
    };
    const unsigned rowbytes = 4u * (unsigned)a.ldx;      // [ldx halves | ldx halves]
    const i4 rs_x = make_rsrc(static_cast<const unsigned char*>(a.x) + (size_t)row0 * rowbytes, (unsigned)min(ROWS, Mlive - row0) * rowbytes);
    const i4 rs_w = make_rsrc(a.wq, (unsigned)(2 * a.n_col_blocks) * NS * 6 * 1024u);
    // this wave's pieces of a chunk: x blocks 4 wave .. + 3 (8 rows each), fragments 3 wave .. + 2
    unsigned xvoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = wave * 4 + j, rp = 4 * b + (lane >> 4), t = (lane & 15) ^ (rp & 15);
        const int row = 2 * rp + (t >> 3), q = t & 7;
        xvoff[j] = (unsigned)row * rowbytes + (unsigned)(q >> 2) * 2u * (unsigned)a.ldx + (unsigned)(q & 3) * 16u;
    }
    const unsigned lds0 = XFH_LDS_ADDR(smem_ld, smem_ld);
    auto issue_x1 = [&](int kc, int xs, int j) { XFH_DMA_B128_TO_LDS(lds0 + xs * XB + (wave * 4 + j) * 1024, xvoff[j], rs_x, kc * 64); };
    auto issue_w1 = [&](int kc, int j) {
        const int f = wave * 3 + j, chf = f / 12, st = (f % 12) / 6, spcb = f % 6;
        XFH_DMA_B128_TO_LDS(lds0 + WOFF + (kc & 1) * WB + f * 1024, lane * 16, rs_w, ((((2 * nbk + chf) * NS + 2 * kc + st) * 6) + spcb) * 1024);
    };
    auto issue_x = [&](int kc, int xs) {
#pragma unroll
        for (int j = 0; j < 4; ++j) issue_x1(kc, xs, j);
    };
    auto issue_w = [&](int kc) {
#pragma unroll
        for (int j = 0; j < 3; ++j) issue_w1(kc, j);
    };
    // operand addresses within a stage: x piece (row, q = 4 plane + 2 step + half), this wave's fragments
    unsigned xoff[2][2][2];      // [row half-block][K step][plane]
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int row = rg * 64 + rb * 32 + l31, base = ((row & 1) << 3) ^ ((row >> 1) & 15);
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) xoff[rb][st][pl] = (unsigned)(row >> 1) * 256u + (unsigned)(base ^ (4 * pl + 2 * st + half)) * 16u;
    }
    const unsigned woff = WOFF + ch * 12 * 1024 + lane * 16;

    f32x16 acc[2][2];
    f16x8 ah[2], al[2], w0[3][2], w1[3][2];
#if XFH_LFXD_TRACE
    long long tstamp[6] = {0, 0, 0, 0, 0, 0};
#endif
    auto mfma4 = [&](f16x8 (&wf)[2], f16x8 (&xo)[2]) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cb], xo[rb], acc[rb][cb], 0, 0, 0);
    };
    auto pin_w = [&](f16x8 (&w)[3][2]) {
#pragma unroll
        for (int sp = 0; sp < 3; ++sp)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) XFH_AGPR(w[sp][cb]);
    };
    // tail: 0 = a chunk of the steady state, 1 = the last but one (no rows left to request), 2 = the last
    auto chunk = [&](int kc, int xs, auto tail) {
        constexpr int TAIL = decltype(tail)::value;
        // The seven pieces this wave requests per chunk stand BETWEEN its MFMA quads, one at a time (trace of the first form, all seven at the chunk's top: waves 4-7 started
        // their MFMAs 1250 counts behind waves 0-3, who then waited as long at the barrier.  Spreading the requests did not change the chunk's time: the pair of waves on a SIMD
        // shares the matrix pipe and the loser of the arbitration is late either way -- DESIGN 3.7; a CU draws 55-60 B/clk by LDS-DMA from L2, this kernel asks for 22).
        // Order: the fragments first (the end-of-chunk wait leaves the four youngest in flight).
        constexpr bool DW = TAIL < 2, DX = TAIL < 1;
        const int xn = xs == 0 ? 2 : xs - 1;      // stage (kc + 2) % 3 = (kc - 1) % 3: read in chunk kc - 1, everybody is past that chunk's barrier (so for the fragments' stage)
#if XFH_LFXD_TRACE
        const bool tr = TAIL == 0 && kc == 8 && blockIdx.x == 0;
        if (tr) tstamp[0] = __builtin_amdgcn_s_memtime();
#endif
        const unsigned char* cur = smem_ld + xs * XB;
        const unsigned char* curw = smem_ld + (kc & 1) * WB;
        auto read_x = [&](f16x8 (&xo)[2], int st, int pl) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) xo[rb] = *reinterpret_cast<const f16x8*>(cur + xoff[rb][st][pl]);
        };
        auto read_w = [&](f16x8 (&w)[3][2], int st) {
#pragma unroll
            for (int sp = 0; sp < 3; ++sp)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) w[sp][cb] = *reinterpret_cast<const f16x8*>(curw + woff + (st * 6 + sp * 2 + cb) * 1024);
        };
        read_x(ah, 0, 0); read_x(al, 0, 1);
        read_w(w0, 0);
        read_w(w1, 1);
        if constexpr (DW) issue_w1(kc + 1, 0);
        pin_w(w0);
        XFH_AGPR(ah[0]); XFH_AGPR(ah[1]); XFH_AGPR(al[0]); XFH_AGPR(al[1]);
#if XFH_LFXD_TRACE
        if (tr) tstamp[1] = __builtin_amdgcn_s_memtime();
#endif
        mfma4(w0[2], ah);
        if constexpr (DW) issue_w1(kc + 1, 1);
        mfma4(w0[0], ah);
        if constexpr (DW) issue_w1(kc + 1, 2);
        read_x(ah, 1, 0);
        mfma4(w0[1], al);
        if constexpr (DX) issue_x1(kc + 2, xn, 0);
        read_x(al, 1, 1);
#if XFH_LFXD_TRACE
        if (tr) tstamp[2] = __builtin_amdgcn_s_memtime();
#endif
        pin_w(w1);
        XFH_AGPR(ah[0]); XFH_AGPR(ah[1]); XFH_AGPR(al[0]); XFH_AGPR(al[1]);
#if XFH_LFXD_TRACE
        if (tr) tstamp[3] = __builtin_amdgcn_s_memtime();
#endif
        mfma4(w1[2], ah);
        if constexpr (DX) issue_x1(kc + 2, xn, 1);
        mfma4(w1[0], ah);
        if constexpr (DX) issue_x1(kc + 2, xn, 2);
        mfma4(w1[1], al);
        if constexpr (DX) issue_x1(kc + 2, xn, 3);
        XFH_SCHED_FENCE();
#if XFH_LFXD_TRACE
        if (tr) tstamp[4] = __builtin_amdgcn_s_memtime();
#endif
        // the next chunk's pieces have landed, for every wave (the four row pieces just requested may still be on their way); everybody has read this chunk
        if constexpr (TAIL == 0) { XFH_WAIT_VMCNT(4); __syncthreads(); }
        if constexpr (TAIL == 1) { XFH_WAIT_VMCNT0(); __syncthreads(); }
#if XFH_LFXD_TRACE
        if (tr) tstamp[5] = __builtin_amdgcn_s_memtime();
#endif
    };
    static_assert(NKC >= 3, "three stages of rows");
    issue_w(0);
    issue_x(0, 0);
    issue_x(1, 1);
    XFH_WAIT_VMCNT(4);
    __syncthreads();
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;
    int xs = 0;
#pragma unroll 1
    for (int kc = 0; kc < NKC - 2; ++kc) { chunk(kc, xs, std::integral_constant<int, 0>{}); xs = xs == 2 ? 0 : xs + 1; }
    chunk(NKC - 2, xs, std::integral_constant<int, 1>{});
    xs = xs == 2 ? 0 : xs + 1;
    chunk(NKC - 1, xs, std::integral_constant<int, 2>{});

    // ---- epilogue (as above), one column half at a time through the stages' LDS
    unsigned amax = 0;
    _Float16* yp = static_cast<_Float16*>(a.y);
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        XFH_WAIT_VMCNT0();    // (tools/check_dma_barriers.py: every barrier of a DMA kernel; here the first pass's stores)
        __syncthreads();      // (the last chunk's reads / the other half's copy)
        if (ch == pass) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c0 = cb * 32 + 8 * g + 4 * half;
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            v[j] = fmaf(acc[rb][cb][4 * g + j], FX_SCALE_INV, a.bias[n0 + 64 * pass + c0 + j]);
                            if (a.relu) v[j] = fmaxf(v[j], 0.f);
                        }
                        uint2 h, l;
                        split2_f16_scalar(v[0], v[1], h.x, l.x);
                        split2_f16_scalar(v[2], v[3], h.y, l.y);
                        fx_track_h(amax, h.x, true); fx_track_h(amax, h.y, true);
                        unsigned char* p = smem_ld + (rg * 64 + rb * 32 + l31) * OUT_RB + 2 * c0;
                        *reinterpret_cast<uint2*>(p) = h;
                        *reinterpret_cast<uint2*>(p + 128) = l;
                    }
        }
        XFH_WAIT_VMCNT0();
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int rl = ps * 32 + (tid >> 4), piece = tid & 15;
            const u4 v = *reinterpret_cast<const u4*>(smem_ld + rl * OUT_RB + 16 * piece);
            if (row0 + rl < Mlive) *reinterpret_cast<u4*>(yp + (long)(row0 + rl) * (2 * a.ldy) + (piece >> 3) * a.ldy + n0 + 64 * pass + 8 * (piece & 7)) = v;
        }
    }
#if XFH_LFXD_TRACE
    if (blockIdx.x == 0 && lane == 0 && a.ldy == 512) {      // (debug: the stamps of chunk 8, per wave, over the first bytes of the output -- tools/gpu_fine_trace.py reads them back)
        unsigned* t = static_cast<unsigned*>(a.y) + wave * 8;
        t[0] = 0x7ace7ace;
        for (int i = 1; i < 6; ++i) t[i] = (unsigned)(tstamp[i] - tstamp[0]);
        t[6] = (unsigned)tstamp[0];
    }
#endif
    fx_report_h(amax, a.status);
}

}  // namespace xfh
