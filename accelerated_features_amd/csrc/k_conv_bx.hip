// 3x3 convolution of the 24-channel layers (block2.0 / block2.1 stride 1, block3.0 stride 2; modules/model.py:56-62) on the fp16 matrix cores in the fp16-pair
// arithmetic (bx_split.hpp): fp32-equivalent product sums from three v_mfma_f32_32x32x16_f16 per K = 16 -- 96 pipe cycles against 512 for the same K on
// v_mfma_f32_32x32x2_f32 -- as a direct convolution: no input / output transforms, no cross-wave exchange, one barrier pair per tile.
//
// Layout: a persistent workgroup (4 waves, two workgroups per CU) walks 8x32-pixel output tiles.
//   * staging: thread (pixel, 8-channel group) loads its 8 raw fp32 values from the NCHW planes (32 loads in flight per thread),
//     splits them and writes two 16-byte rows (high parts, low parts) into LDS:  [pixel of the 10x34 halo tile][fragment][24 channels] fp16 (112 B per pixel:
//     an odd multiple of 16 B, so the 16 lanes of a ds_read_b128 group hit 16 distinct 16-byte slots);
//   * GEMM: M = cout (A = weights), N = 32 pixels of one output row (B), K = (tap, channel) in 8-channel groups, two groups per
//     K=16 step (lane half 0 / 1), 27 groups + 1 zero group = 14 steps.  The q0 weight fragments of ALL steps live in registers in operand
//     order (loaded once per workgroup), q1 and q2 are read from LDS once per step; a wave owns two output rows (two accumulators), reads two 16-byte B rows per
//     row and step and issues 3 MFMAs on them;
//   * D: lane (pixel, half) holds couts (r&3) + 8 (r>>2) + 4 half: bias, ReLU, one coalesced 128-byte store per cout and half-wave.
#include "kernels.hpp"
#include "bx_split.hpp"
#include <cstdlib>
#include <type_traits>

namespace xfh {

struct BxArgs {
    const float* in;
    const uint4* wfrag;        // [step][fragment q0, q1, q2][64 lanes] 8 fp16 each (api.hip)
    const float* bias;
    float* out;
    int relu, H, W, B, tiles_x, tiles;
    int lag;                   // first-tile delay of the second workgroup of a CU, in units of 512 cycles
    long long* trace;          // debug: 6 s_memtime stamps per tile, 10 tiles, per workgroup (NULL in production)
    int cold;
    int* status;               // range guard (bx_split.hpp), may be NULL
};

template <int CIN, int COUT>
struct BxCfg {
    static constexpr int TH = 8, TW = 32, IH = TH + 2, IW = TW + 2, NPIX = IH * IW;
    static constexpr int NXS = 2;      // input fragments per pixel (high parts, low parts)
    // bytes per pixel / per fragment row; an ODD multiple of 16 B per pixel keeps the 16 lanes of a ds_read_b128 group on distinct banks (2 x 48 + 16 of padding)
    static constexpr int CG = CIN / 8, SPLB = CIN * 2, PIXB = ((NXS * SPLB / 16) | 1) * 16;
    static constexpr int KG = 9 * CG, NSTEP = (KG + 1) / 2;
    static constexpr int NITEM = NPIX * CG, NIT = (NITEM + 255) / 256;
    static constexpr bool WM_LDS = true;
    static constexpr int TILE_BYTES = NPIX * PIXB, WL_BYTES = NSTEP * 64 * 16;
    static constexpr int BIAS_OFF = TILE_BYTES + (WM_LDS ? 2 : 1) * WL_BYTES, LDS_BYTES = BIAS_OFF + 32 * 4;
    static_assert(CIN % 8 == 0 && COUT <= 32, "one cout block, 8-channel groups");
    // byte offset of K group kg (tap, channel group) relative to the lane's own pixel
    static constexpr int koff(int kg) {
        const int g = kg < KG ? kg : KG - 1;       // the zero group re-reads the last one (finite values x zero weights)
        const int tap = g / CG, cg = g % CG;
        return ((tap / 3) * IW + tap % 3) * PIXB + cg * 16;
    }
};

template <int CIN, int COUT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_bx_kernel(BxArgs a) {
    kernel_entry_hooks(a.cold);      // debug: code-position shift / cold instruction cache (common.hpp)
    using Cfg = BxCfg<CIN, COUT>;
    constexpr int NXS = Cfg::NXS;
    using frag_t = f16x8;
    auto mfma = [](frag_t x, frag_t y, f32x16 c) __attribute__((always_inline)) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0); };
    constexpr int TH = Cfg::TH, TW = Cfg::TW, IW = Cfg::IW, NPIX = Cfg::NPIX, CG = Cfg::CG, PIXB = Cfg::PIXB, SPLB = Cfg::SPLB;
    constexpr int NSTEP = Cfg::NSTEP, NIT = Cfg::NIT;
    constexpr bool WM_LDS = Cfg::WM_LDS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_bx[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t HW = (size_t)a.H * a.W;

    // weight fragments in operand order: lane (cout l31, half) holds K group 2 s + half of every step.  wh (= q0) stays in registers for the whole kernel;
    // wm (= q1) and wl (= q2) are read from LDS once per step: all three in registers (168)
    // leave hipcc two fragment buffers and an lgkmcnt(0) in front of every MFMA group, and no room for the prefetched next tile
    frag_t wf[NSTEP][WM_LDS ? 1 : 2];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s)
#pragma unroll
        for (int q = 0; q < (WM_LDS ? 1 : 2); ++q) wf[s][q] = __builtin_bit_cast(frag_t, a.wfrag[(s * 3 + q) * 64 + lane]);
    unsigned char* wl_lds = smem_bx + Cfg::TILE_BYTES;            // [step][lane] 16 B
    unsigned char* wm_lds = wl_lds + Cfg::WL_BYTES;
    for (int s = wave; s < NSTEP; s += 4) {
        *reinterpret_cast<uint4*>(wl_lds + (s * 64 + lane) * 16) = a.wfrag[(s * 3 + 2) * 64 + lane];
        if (WM_LDS) *reinterpret_cast<uint4*>(wm_lds + (s * 64 + lane) * 16) = a.wfrag[(s * 3 + 1) * 64 + lane];
    }
    // pin the wait for the weight loads here: hipcc otherwise keeps an s_waitcnt vmcnt(32 + k) in front of each step's first use INSIDE
    // the tile loop, where the counter also holds the previous tile's stores (the MFMA loop would wait for their write acks)
#pragma unroll
    for (int s = 0; s < NSTEP; ++s)
#pragma unroll
        for (int q = 0; q < (WM_LDS ? 1 : 2); ++q) asm volatile("" : "+v"(wf[s][q]));
    // the bias lives in LDS too: a global load in the epilogue would wait (vmcnt counts in order) for the stores issued before it
    float* bias_lds = reinterpret_cast<float*>(smem_bx + Cfg::BIAS_OFF);
    if (tid < 32) bias_lds[tid] = a.bias[tid];

    // staging items of this thread: item = cg * NPIX + pixel -> (tile row, tile column, LDS byte offset); cg >= CG: no item
    int it_rc[NIT], it_lds[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int item = tid + 256 * i;
        const int cg = item / NPIX, pix = item - cg * NPIX;
        const int r = pix / IW, c = pix - r * IW;
        it_rc[i] = cg < CG ? (cg << 16) | (r << 8) | c : -1;
        it_lds[i] = pix * PIXB + cg * 16;
    }
    const int lane_off = (2 * wave * IW + l31) * PIXB;
    const int total = a.tiles * a.B;
    long long* tr = a.trace && tid == 0 ? a.trace + (size_t)blockIdx.x * 64 : nullptr;
    int tix = 0;
#define BX_STAMP(k) { if (tr && tix < 10) tr[tix * 6 + (k)] = __builtin_amdgcn_s_memtime(); }

    auto tile_of = [&](int vid, int& b, int& oy0, int& ox0) {
        int tile;
        xcd_group_map(vid, a.tiles, a.B, b, tile);          // vid < tiles * B: never a padding id
        const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
        oy0 = tyi * TH; ox0 = txi * TW;
    };
    // raw fp32 values of a tile: 8 channels of one pixel per item, all loads of a thread in flight together; out-of-image pixels
    // carry an out-of-range offset (the buffer load returns the zero padding)
    float v[NIT][8];
    unsigned amax = 0;                        // the largest fp16 high parts converted (range guard: bx_split.hpp)
    auto issue_loads = [&](int vid) {
        int b, oy0, ox0;
        tile_of(vid, b, oy0, ox0);
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)b * CIN * HW), 0, (int)(CIN * HW * sizeof(float)), 0x00020000);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int cg = it_rc[i] >> 16, gy = oy0 - 1 + ((it_rc[i] >> 8) & 0xff), gx = ox0 - 1 + (it_rc[i] & 0xff);
            const bool ok = it_rc[i] >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            const int voff = ok ? (int)((((size_t)cg * 8) * HW + (size_t)gy * a.W + gx) * 4) : (int)0x80000000;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[i][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_in, voff, (int)(k * HW * 4), 0));
        }
    };
    auto stage_write = [&]() {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            uint4 h, l;
            split2_f16(v[i][0], v[i][1], h.x, l.x); split2_f16(v[i][2], v[i][3], h.y, l.y);
            split2_f16(v[i][4], v[i][5], h.z, l.z); split2_f16(v[i][6], v[i][7], h.w, l.w);
            fx_track_h(amax, h.x, true); fx_track_h(amax, h.y, true); fx_track_h(amax, h.z, true); fx_track_h(amax, h.w, true);      // (on the high parts: bx_split.hpp)
            if (it_rc[i] >= 0) {
                unsigned char* p = smem_bx + it_lds[i];
                *reinterpret_cast<uint4*>(p) = h;
                *reinterpret_cast<uint4*>(p + SPLB) = l;
            }
        }
    };

    int vid = blockIdx.x;
    if (vid >= total) return;
    // (Also tried: ONE 8-wave workgroup per CU whose two halves own different tiles / LDS buffers and swap roles at a barrier per slot --
    // one half on the matrix pipe while the other stores and stages.  102 us against 93: a lone MFMA wave per SIMD has nobody to cover
    // its LDS latency, a slot took 10.8 k cycles instead of the 5.5 k of its 168 MFMAs.)
    // Two workgroups share a CU and keep whatever phase lag they start with (a lag x between their MFMA phases is preserved from
    // tile to tile: period = 2 M + O - x for M cycles of MFMA and O cycles of staging + stores per tile).  Launched together they
    // run in phase: both on the matrix pipe, then both off it.  The workgroup that was allocated second on its CU (LDS base != 0)
    // starts half a period late, so one stages / stores while the other one multiplies.
    {
        unsigned lds_alloc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(lds_alloc));
        if (tr) tr[62] = lds_alloc;
        if ((lds_alloc & 0xffu) != 0)
            for (int i = 0; i < a.lag; ++i) __builtin_amdgcn_s_sleep(8);
    }
    issue_loads(vid);
    for (;;) {
        int b, oy0, ox0;
        tile_of(vid, b, oy0, ox0);
        BX_STAMP(0)
        stage_write();
        BX_STAMP(1)
        __syncthreads();
        BX_STAMP(2)
        const int nvid = vid + (int)gridDim.x;
        if (nvid < total) issue_loads(nvid);          // the next tile's loads fly under this tile's MFMAs
        // ---- 14 K steps x 2 rows x 3 MFMAs ---------------------------------------------------------------------------
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        // operands of step s+1 are read while the MFMAs of step s issue (two register sets, pinned: left alone hipcc sinks the reads
        // in front of their first use).  The K groups of the two lane halves differ by one of three byte deltas -> three base registers.
        struct Frag { frag_t x[2][NXS]; frag_t wl, wm; };
        Frag f[2];
        auto load = [&](int s, Frag& o) {
            const int k0 = Cfg::koff(2 * s), dk = Cfg::koff(2 * s + 1) - k0;
            const unsigned char* p = smem_bx + (lane_off + half * dk) + k0;
            o.wl = *reinterpret_cast<const frag_t*>(wl_lds + (s * 64 + lane) * 16);
            if (WM_LDS) o.wm = *reinterpret_cast<const frag_t*>(wm_lds + (s * 64 + lane) * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < NXS; ++q) o.x[j][q] = *reinterpret_cast<const frag_t*>(p + j * IW * PIXB + q * SPLB);
        };
        load(0, f[0]);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const Frag& c = f[s & 1];
            if (s + 1 < NSTEP) load(s + 1, f[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const frag_t wh = wf[s][0], wm = WM_LDS ? c.wm : wf[s][WM_LDS ? 0 : 1];
            // small terms first; the two rows alternate (independent accumulators)
#define BX_MM(A, Q) { acc[0] = mfma(A, c.x[0][Q], acc[0]); acc[1] = mfma(A, c.x[1][Q], acc[1]); }
            BX_MM(c.wl, 0) BX_MM(wm, 1) BX_MM(wh, 0)      // fragments (2^11 w - q0, w, q0 = fp16(2^11 w)) x (xh, xl, xh): all at scale 2^11
#undef BX_MM
            __builtin_amdgcn_sched_barrier(0);
        }
        // idle slots: the epilogue's VALU code must not land in operand registers of the last MFMAs.  Tied to the accumulators: an asm without
        // operands is no anchor -- hipcc moved it in front of the fx form's last two MFMAs and the epilogue's address arithmetic behind them (ISA audit)
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
        __builtin_amdgcn_sched_barrier(0);
        BX_STAMP(3)
        // ---- bias, ReLU, store ------------------------------------------------------------------------------------------
        const int ox = ox0 + l31;
        float bs[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 t = *reinterpret_cast<const float4*>(bias_lds + 8 * g4 + 4 * half);
            bs[4 * g4] = t.x; bs[4 * g4 + 1] = t.y; bs[4 * g4 + 2] = t.z; bs[4 * g4 + 3] = t.w;
        }
        // buffer stores: one address dword per lane instead of two (a store's issue time is its VGPR traffic); lanes outside the image
        // carry an out-of-range offset and are dropped
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)b * COUT * HW), 0, (int)(COUT * HW * sizeof(float)), 0x00020000);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int oy = oy0 + 2 * wave + j;
            const int voff = oy < a.H && ox < a.W ? (int)(((size_t)(4 * half) * HW + (size_t)oy * a.W + ox) * 4) : (int)0x80000000;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co0 = (r & 3) + 8 * (r >> 2);          // cout of lane half 0; half 1 holds co0 + 4
                if (co0 < COUT) {                                // compile-time (r < 12 for 24 channels)
                    float y = fmaf(acc[j][r], FX_SCALE_INV, bs[r]);
                    if (a.relu) y = fmaxf(y, 0.f);
                    const bool okc = COUT % 8 == 0 || co0 + 4 * half < COUT;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rs_out, okc ? voff : (int)0x80000000, (int)(co0 * HW * 4), 0);
                }
            }
        }
        BX_STAMP(4)
        if (nvid >= total) break;
        __syncthreads();         // every wave is done with the tile before the next one is staged
        BX_STAMP(5)
        ++tix;
        vid = nvid;
    }
    fx_report_h(amax, a.status);
#undef BX_STAMP
}

// ------------------------------------------------------------------------------------------------------------------------------
// Round 6: conv_bxd_kernel -- the same convolution with the STAGING INSIDE THE MFMA STREAM.  The counters said what conv_bx_kernel does with its time: matrix
// pipe busy 38 % of its cycles, vector ALU 21 % -- it waits.  A workgroup there stages a tile, meets at a barrier, multiplies, meets again; the second workgroup of
// the CU is meant to fill the gaps and does so for a third of them.  Here ONE workgroup of eight waves (two per SIMD) owns the CU and all of its LDS:
//   * tiles of 16 x 32 outputs (halo 18 x 34: 1.20 x the pixels instead of 1.33), TWO tile buffers: while tile t is multiplied out of one, tile t + 1 is
//     split and written into the other BY THE SAME WAVES, in the slots between the K steps (a staging item = 8 channels of a pixel: 8 conversions + 2 LDS writes, one item
//     per slot in four of the fourteen steps; a wave's vector instructions beside its own MFMAs are free up to ~ 4 per MFMA), and the raw values of tile t + 2 are
//     requested into the registers an item has just left -- a whole tile ahead of their use;
//   * ONE barrier per tile (everybody is done reading buffer t and writing buffer t + 1), no staging phase, no phase lag to tune;
//   * fp16(w) is not kept in LDS: q1 = 2^-11 q0 (exact scaling of the same significand, the block1 kernels' trick): four v_pk_mul_f16 per step buy the 14 KB the
//     second tile buffer needs (2 x 68 544 + 14 336 + 128 = 151 552 bytes of LDS).
// A wave owns output rows 2 w, 2 w + 1 of the tile (two accumulators), as in conv_bx_kernel; weights q0 of all steps in registers, q2 read from LDS per step.
// ------------------------------------------------------------------------------------------------------------------------------
template <int CIN, int COUT>
struct BxdCfg {
    static constexpr int TH = 16, TW = 32, IH = TH + 2, IW = TW + 2, NPIX = IH * IW;
    static constexpr int NXS = 2;
    static constexpr int CG = CIN / 8, SPLB = CIN * 2, PIXB = ((NXS * SPLB / 16) | 1) * 16;
    static constexpr int KG = 9 * CG, NSTEP = (KG + 1) / 2;
    static constexpr int NITEM = NPIX * CG, NIT = (NITEM + 511) / 512;
    static constexpr int TILE_BYTES = NPIX * PIXB, WL_BYTES = NSTEP * 64 * 16;
    static constexpr int WL_OFF = 2 * TILE_BYTES, BIAS_OFF = WL_OFF + WL_BYTES, LDS_BYTES = BIAS_OFF + 32 * 4;
    static_assert(CIN % 8 == 0 && COUT <= 32 && NIT <= NSTEP / 3 && LDS_BYTES <= 160 * 1024, "one cout block, 8-channel groups, an item per three steps, one CU's LDS");
    static constexpr int koff(int kg) {
        const int g = kg < KG ? kg : KG - 1;
        const int tap = g / CG, cg = g % CG;
        return ((tap / 3) * IW + tap % 3) * PIXB + cg * 16;
    }
};

// IN_CL / OUT_CL: the input / output is channels-last ((B, H, W, C): the links block2.0 -> block2.1 -> block3.0 inside the backbone).  A memory instruction costs its issue time
// whatever else the SIMD does: an item's 8 channels are then two 16-byte loads instead of eight 4-byte ones, a lane's 4 consecutive couts one 16-byte store instead of four.
template <int CIN, int COUT, bool IN_CL = false, bool OUT_CL = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_bxd_kernel(BxArgs a) {
    kernel_entry_hooks(a.cold);      // debug: code-position shift / cold instruction cache (common.hpp)
    using Cfg = BxdCfg<CIN, COUT>;
    constexpr int NXS = Cfg::NXS;
    using frag_t = f16x8;
    auto mfma = [](frag_t x, frag_t y, f32x16 c) __attribute__((always_inline)) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0); };
    constexpr int TH = Cfg::TH, TW = Cfg::TW, IW = Cfg::IW, NPIX = Cfg::NPIX, PIXB = Cfg::PIXB, SPLB = Cfg::SPLB;
    constexpr int NSTEP = Cfg::NSTEP, NIT = Cfg::NIT, TILE_BYTES = Cfg::TILE_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_bx[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t HW = (size_t)a.H * a.W;

    // q0 = fp16(2^11 w) of every step in registers (operand order: lane (cout l31, half) holds K group 2 s + half); q2 in LDS
    frag_t wf[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) wf[s] = __builtin_bit_cast(frag_t, a.wfrag[(s * 3 + 0) * 64 + lane]);
    unsigned char* wl_lds = smem_bx + Cfg::WL_OFF;            // [step][lane] 16 B
    for (int s = wave; s < NSTEP; s += 8) *reinterpret_cast<uint4*>(wl_lds + (s * 64 + lane) * 16) = a.wfrag[(s * 3 + 2) * 64 + lane];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) asm volatile("" : "+v"(wf[s]));      // the wait for the weight loads belongs here, not into the tile loop
    float* bias_lds = reinterpret_cast<float*>(smem_bx + Cfg::BIAS_OFF);
    if (tid < 32) bias_lds[tid] = a.bias[tid];

    // staging items of this thread: item = cg * NPIX + pixel -> (tile row, tile column, LDS byte offset).  4 x 512 slots for 1836 items: the threads behind the last item
    // take an item a second time (the same values to the same place: no branch, no exec mask in the MFMA stream)
    int it_rc[NIT], it_lds[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        int item = tid + 512 * i;
        item = item >= Cfg::NITEM ? item - Cfg::NITEM : item;
        // planes: consecutive lanes = consecutive pixels of a channel group; channels-last: consecutive lanes = the channel groups of a pixel, then the next pixel (32 contiguous bytes each)
        const int cg = IN_CL ? item % Cfg::CG : item / NPIX, pix = IN_CL ? item / Cfg::CG : item - cg * NPIX;
        const int r = pix / IW, c = pix - r * IW;
        it_rc[i] = (cg << 16) | (r << 8) | c;
        it_lds[i] = pix * PIXB + cg * 16;
    }
    const int lane_off = (2 * wave * IW + l31) * PIXB;
    const int total = a.tiles * a.B;

    auto tile_of = [&](int vid, int& b, int& oy0, int& ox0) {
        int tile;
        xcd_group_map(vid, a.tiles, a.B, b, tile);
        const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
        oy0 = tyi * TH; ox0 = txi * TW;
    };
    unsigned amax = 0;                        // the largest fp16 high parts converted (range guard: bx_split.hpp)
    // where a tile's raw values come from (vid >= total: every offset out of range -- the loads return zeros that are staged into a buffer nobody reads; no branch)
    struct TileSrc { __amdgpu_buffer_rsrc_t rs; int oy0, ox0, live; };
    auto src_of = [&](int vid) {
        TileSrc t;
        int b = 0, oy0 = 0, ox0 = 0;
        t.live = vid < total;
        if (t.live) tile_of(vid, b, oy0, ox0);
        t.oy0 = oy0; t.ox0 = ox0;
        t.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)b * CIN * HW), 0, (int)(CIN * HW * sizeof(float)), 0x00020000);
        return t;
    };
    auto issue_item = [&](const TileSrc& t, int i, float (&v)[8]) __attribute__((always_inline)) {
        const int cg = it_rc[i] >> 16, gy = t.oy0 - 1 + ((it_rc[i] >> 8) & 0xff), gx = t.ox0 - 1 + (it_rc[i] & 0xff);
        const bool ok = (bool)((int)(t.live != 0) & (int)((unsigned)gy < (unsigned)a.H) & (int)((unsigned)gx < (unsigned)a.W)); 
        if constexpr (IN_CL) {
            const int voff = ok ? (int)((((size_t)gy * a.W + gx) * CIN + cg * 8) * 4) : (int)0x80000000;
            const uint4 lo = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(t.rs, voff, 0, 0)), hi = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(t.rs, voff, 16, 0));
            v[0] = __uint_as_float(lo.x); v[1] = __uint_as_float(lo.y); v[2] = __uint_as_float(lo.z); v[3] = __uint_as_float(lo.w);
            v[4] = __uint_as_float(hi.x); v[5] = __uint_as_float(hi.y); v[6] = __uint_as_float(hi.z); v[7] = __uint_as_float(hi.w);
        } else {
            const int voff = ok ? (int)((((size_t)cg * 8) * HW + (size_t)gy * a.W + gx) * 4) : (int)0x80000000;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(t.rs, voff, (int)(k * HW * 4), 0));
        }
    };
    auto stage_item = [&](unsigned char* buf, int i, const float (&v)[8]) __attribute__((always_inline)) {
        uint4 h, l;
        split2_f16(v[0], v[1], h.x, l.x); split2_f16(v[2], v[3], h.y, l.y);
        split2_f16(v[4], v[5], h.z, l.z); split2_f16(v[6], v[7], h.w, l.w);
        fx_track_h(amax, h.x, true); fx_track_h(amax, h.y, true); fx_track_h(amax, h.z, true); fx_track_h(amax, h.w, true);
        unsigned char* p = buf + it_lds[i];
        *reinterpret_cast<uint4*>(p) = h;
        *reinterpret_cast<uint4*>(p + SPLB) = l;
    };
    static_assert(NIT == 4, "four staging items per thread, spelled out below");

    int vid = blockIdx.x;
    if (vid >= total) return;
    float va[8], vb[8];                       // raw fp32 values of two staging items in flight
    // prologue: tile 0 into buffer 0 with every pipe idle
    {
        const TileSrc t0 = src_of(vid);
        issue_item(t0, 0, va); issue_item(t0, 1, vb);
        stage_item(smem_bx, 0, va); issue_item(t0, 2, va);
        stage_item(smem_bx, 1, vb); issue_item(t0, 3, vb);
        stage_item(smem_bx, 2, va); stage_item(smem_bx, 3, vb);
    }
    __syncthreads();
    int cur = 0;
    for (;;) {
        int b, oy0, ox0;
        tile_of(vid, b, oy0, ox0);
        const unsigned char* tb = smem_bx + cur * TILE_BYTES;            // this tile
        unsigned char* nb = smem_bx + (cur ^ 1) * TILE_BYTES;           // the next one is requested AND staged here, inside this tile's MFMAs
        const TileSrc t1 = src_of(vid + (int)gridDim.x);
        // ---- 14 K steps x 2 rows x 3 MFMAs; in their slots the next tile's four items, two in flight: requested at steps 0 / 3 / 6 / 9, split and written at 5 / 8 / 11 / 13
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        struct Frag { frag_t x[2][NXS]; frag_t wl; };
        Frag f[2];
        // A step's operand address = tile + lane's pixel + (upper half-wave: the distance dk to the step's second K group) + a constant.  dk takes few values (16 within a tap,
        // else the jump to the next tap / tap row): their bases are computed ONCE per tile, in front of its first MFMA.  Computed per step, the addition was the first vector
        // instruction behind a step's MFMAs, and hipcc gave it a register the last of them had just read as an operand (tools/check_mfma_war.py).
        struct DkTab { int n; int dk[8]; int ix[NSTEP]; };
        constexpr DkTab DK = [] {
            DkTab t{};
            for (int s = 0; s < NSTEP; ++s) {
                const int dk = Cfg::koff(2 * s + 1) - Cfg::koff(2 * s);
                int ix = -1;
                for (int k = 0; k < t.n; ++k) if (t.dk[k] == dk) ix = k;
                if (ix < 0) { ix = t.n; if (t.n < 8) t.dk[t.n] = dk; ++t.n; }
                t.ix[s] = ix;
            }
            return t;
        }();
        static_assert(DK.n <= 4, "a handful of distinct half-wave distances");
        unsigned pbase[DK.n];
#pragma unroll
        for (int k = 0; k < DK.n; ++k) { pbase[k] = (unsigned)(cur * TILE_BYTES + lane_off + half * DK.dk[k]); asm volatile("" : "+v"(pbase[k])); }
        auto load = [&](int s, Frag& o) __attribute__((always_inline)) {
            const int k0 = Cfg::koff(2 * s);
            const unsigned char* p = smem_bx + pbase[DK.ix[s]] + k0;
            o.wl = *reinterpret_cast<const frag_t*>(wl_lds + (s * 64 + lane) * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < NXS; ++q) o.x[j][q] = *reinterpret_cast<const frag_t*>(p + j * IW * PIXB + q * SPLB);
        };
        load(0, f[0]);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const Frag& c = f[s & 1];
            if (s + 1 < NSTEP) load(s + 1, f[(s + 1) & 1]);
            const frag_t wh = wf[s];
            frag_t wm = wh * (_Float16)0.00048828125f;            // fp16(w) = 2^-11 fp16(2^11 w)
            asm volatile("" : "+v"(wm));                           // (computed HERE, in front of the MFMA group)
            __builtin_amdgcn_sched_barrier(0);
#define BX_MM(A, Q) { acc[0] = mfma(A, c.x[0][Q], acc[0]); acc[1] = mfma(A, c.x[1][Q], acc[1]); }
            BX_MM(c.wl, 0) BX_MM(wm, 1) BX_MM(wh, 0)      // fragments (2^11 w - q0, w, q0 = fp16(2^11 w)) x (xh, xl, xh): all at scale 2^11
#undef BX_MM
            __builtin_amdgcn_sched_barrier(0);
            if (s == 0) issue_item(t1, 0, va);
            if (s == 3) issue_item(t1, 1, vb);
            if (s == 5) stage_item(nb, 0, va);
            if (s == 6) issue_item(t1, 2, va);
            if (s == 8) stage_item(nb, 1, vb);
            if (s == 9) issue_item(t1, 3, vb);
            if (s == 11) stage_item(nb, 2, va);
            if (s == 13) stage_item(nb, 3, vb);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));      // idle slots: the epilogue's vector code must not land in operands of the last MFMAs
        __builtin_amdgcn_sched_barrier(0);
        // ---- bias, ReLU, store ------------------------------------------------------------------------------------------
        const int ox = ox0 + l31;
        float bs[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 t = *reinterpret_cast<const float4*>(bias_lds + 8 * g4 + 4 * half);
            bs[4 * g4] = t.x; bs[4 * g4 + 1] = t.y; bs[4 * g4 + 2] = t.z; bs[4 * g4 + 3] = t.w;
        }
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)b * COUT * HW), 0, (int)(COUT * HW * sizeof(float)), 0x00020000);
        if constexpr (OUT_CL) {      // channels-last: couts (r & 3) + 8 g + 4 half, r & 3 = 0..3 are 16 consecutive bytes of pixel (oy, ox)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int oy = oy0 + 2 * wave + j;
                const int voff = oy < a.H && ox < a.W ? (int)((((size_t)oy * a.W + ox) * COUT + 4 * half) * 4) : (int)0x80000000;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (8 * g < COUT) {
                        float y4[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            y4[q] = fmaf(acc[j][4 * g + q], FX_SCALE_INV, bs[4 * g + q]);
                            if (a.relu) y4[q] = fmaxf(y4[q], 0.f);
                        }
                        const bool okc = COUT % 8 == 0 || 8 * g + 4 * half < COUT;
                        typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
                        const u32x4s u = {__float_as_uint(y4[0]), __float_as_uint(y4[1]), __float_as_uint(y4[2]), __float_as_uint(y4[3])};
                        __builtin_amdgcn_raw_buffer_store_b128(u, rs_out, okc ? voff : (int)0x80000000, (int)(8 * g * 4), 0);
                    }
                }
            }
        } else
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int oy = oy0 + 2 * wave + j;
            const int voff = oy < a.H && ox < a.W ? (int)(((size_t)(4 * half) * HW + (size_t)oy * a.W + ox) * 4) : (int)0x80000000;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co0 = (r & 3) + 8 * (r >> 2);          // cout of lane half 0; half 1 holds co0 + 4
                if (co0 < COUT) {
                    float y = fmaf(acc[j][r], FX_SCALE_INV, bs[r]);
                    if (a.relu) y = fmaxf(y, 0.f);
                    const bool okc = COUT % 8 == 0 || co0 + 4 * half < COUT;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rs_out, okc ? voff : (int)0x80000000, (int)(co0 * HW * 4), 0);
                }
            }
        }
        const int nvid = vid + (int)gridDim.x;
        if (nvid >= total) break;
        __syncthreads();         // every wave has read this tile and written its part of the next one
        vid = nvid;
        cur ^= 1;
    }
    fx_report_h(amax, a.status);
}

// ------------------------------------------------------------------------------------------------------------------------------
// The stride-2 sibling: block3.0 (3x3 / s2, 24 -> 64; modules/model.py:62) on the same staging.  The 10x34 halo tile around an 8x32
// block of INPUT pixels feeds a 4x16 block of outputs (one 32-pixel MFMA block per two output rows); wave (pb, cb) owns pixel block pb
// and cout block cb.  The tile is stored with even and odd columns apart ([row][column parity][column / 2], 2496 B per parity row: a
// multiple of 64 B, so that the second row of a pixel block lands on the same banks as the first): the lanes of a fragment read step by
// two input pixels and would otherwise collide pairwise.  wh and wm of the wave's cout block live in registers, wl in LDS.
// ------------------------------------------------------------------------------------------------------------------------------
template <int CIN>
struct BxS2Cfg {
    static constexpr int NXS = 2;
    static constexpr int IH = 10, IW = 34, NPIX = IH * IW, CG = CIN / 8, SPLB = CIN * 2, PIXB = ((NXS * SPLB / 16) | 1) * 16;
    static constexpr int ROWQ = ((IW / 2) * PIXB + 63) / 64 * 64;          // bytes per (row, column parity)
    static constexpr int KG = 9 * CG, NSTEP = (KG + 1) / 2;
    static constexpr int NIT = (NPIX * CG + 255) / 256;
    static constexpr int TILE_BYTES = IH * 2 * ROWQ, WL_BYTES = 2 * NSTEP * 64 * 16;
    static constexpr int BIAS_OFF = TILE_BYTES + WL_BYTES, LDS_BYTES = BIAS_OFF + 64 * 4;
    static_assert((4 * ROWQ) % 256 == 0, "two output rows = four parity rows must be a multiple of the 64 banks");
    static constexpr int koff(int kg) {
        const int g = kg < KG ? kg : KG - 1;
        const int tap = g / CG, cg = g % CG, dy = tap / 3, dx = tap % 3;
        return (dy * 2 + (dx & 1)) * ROWQ + (dx >> 1) * PIXB + cg * 16;
    }
};

struct BxS2Args {
    const float* in;
    const uint4* wfrag;        // [cout block 2][step][fragment][64 lanes] 8 fp16 each
    const float* bias;
    float* out;
    int relu, H, W, Ho, Wo, B, tiles_x, tiles;
    int cold;
    int* status;
};

template <int CIN, bool IN_CL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_bxs2_kernel(BxS2Args a) {
    kernel_entry_hooks(a.cold);      // debug: code-position shift / cold instruction cache (common.hpp)
    using Cfg = BxS2Cfg<CIN>;
    constexpr int NXS = Cfg::NXS;
    using frag_t = f16x8;
    auto mfma = [](frag_t x, frag_t y, f32x16 c) __attribute__((always_inline)) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c, 0, 0, 0); };
    constexpr int IW = Cfg::IW, NPIX = Cfg::NPIX, CG = Cfg::CG, PIXB = Cfg::PIXB, SPLB = Cfg::SPLB, ROWQ = Cfg::ROWQ;
    constexpr int NSTEP = Cfg::NSTEP, NIT = Cfg::NIT, COUT = 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_bx[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pb = wave >> 1, cb = wave & 1;
    const size_t HW = (size_t)a.H * a.W, HWo = (size_t)a.Ho * a.Wo;

    frag_t wf[NSTEP][2];                       // q0 = fp16(2^11 w), q1 = fp16(w) of this wave's cout block
#pragma unroll
    for (int s = 0; s < NSTEP; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q) wf[s][q] = __builtin_bit_cast(frag_t, a.wfrag[((cb * NSTEP + s) * 3 + q) * 64 + lane]);
    unsigned char* wl_lds = smem_bx + Cfg::TILE_BYTES;            // [cout block][step][lane] 16 B
    for (int j = wave; j < 2 * NSTEP; j += 4)
        *reinterpret_cast<uint4*>(wl_lds + (j * 64 + lane) * 16) = a.wfrag[(((j / NSTEP) * NSTEP + j % NSTEP) * 3 + 2) * 64 + lane];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q) asm volatile("" : "+v"(wf[s][q]));      // the wait for the weight loads belongs here, not into the tile loop
    float* bias_lds = reinterpret_cast<float*>(smem_bx + Cfg::BIAS_OFF);
    if (tid < 64) bias_lds[tid] = a.bias[tid];

    int it_rc[NIT], it_lds[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int item = tid + 256 * i;
        const int cg = IN_CL ? (item < NPIX * CG ? item % CG : CG) : item / NPIX, pix = IN_CL ? (item / CG) % NPIX : item - cg * NPIX;      // (channels-last: the channel groups of a pixel side by side)
        const int r = pix / IW, c = pix - r * IW;
        it_rc[i] = cg < CG ? (cg << 16) | (r << 8) | c : -1;
        it_lds[i] = (r * 2 + (c & 1)) * ROWQ + (c >> 1) * PIXB + cg * 16;
    }
    const int orow = 2 * pb + (l31 >> 4), ocol = l31 & 15;
    const int lane_off = 4 * orow * ROWQ + ocol * PIXB;
    // this lane's wl fragments: ONE address register, the step in the instruction's offset field (an address computed per step is a VALU
    // write that hipcc put into a fragment register the step's last MFMA had just read)
    const unsigned char* wl_lane = wl_lds + (cb * NSTEP * 64 + lane) * 16;
    const int total = a.tiles * a.B;

    auto tile_of = [&](int vid, int& b, int& iy0, int& ix0) {
        int tile;
        xcd_group_map(vid, a.tiles, a.B, b, tile);
        const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
        iy0 = tyi * 8; ix0 = txi * 32;
    };
    float v[NIT][8];
    unsigned amax = 0;                        // the largest fp16 high parts converted (range guard: bx_split.hpp)
    auto issue_loads = [&](int vid) __attribute__((always_inline)) {
        int b, iy0, ix0;
        tile_of(vid, b, iy0, ix0);
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)b * CIN * HW), 0, (int)(CIN * HW * sizeof(float)), 0x00020000);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int cg = it_rc[i] >> 16, gy = iy0 - 1 + ((it_rc[i] >> 8) & 0xff), gx = ix0 - 1 + (it_rc[i] & 0xff);
            const bool ok = it_rc[i] >= 0 && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            if constexpr (IN_CL) {
                const int voff = ok ? (int)((((size_t)gy * a.W + gx) * CIN + cg * 8) * 4) : (int)0x80000000;
                const uint4 lo = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, 0, 0)), hi = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, 16, 0));
                v[i][0] = __uint_as_float(lo.x); v[i][1] = __uint_as_float(lo.y); v[i][2] = __uint_as_float(lo.z); v[i][3] = __uint_as_float(lo.w);
                v[i][4] = __uint_as_float(hi.x); v[i][5] = __uint_as_float(hi.y); v[i][6] = __uint_as_float(hi.z); v[i][7] = __uint_as_float(hi.w);
            } else {
                const int voff = ok ? (int)((((size_t)cg * 8) * HW + (size_t)gy * a.W + gx) * 4) : (int)0x80000000;
#pragma unroll
                for (int k = 0; k < 8; ++k) v[i][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_in, voff, (int)(k * HW * 4), 0));
            }
        }
    };
    auto stage_write = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            uint4 h, l;
            split2_f16(v[i][0], v[i][1], h.x, l.x); split2_f16(v[i][2], v[i][3], h.y, l.y);
            split2_f16(v[i][4], v[i][5], h.z, l.z); split2_f16(v[i][6], v[i][7], h.w, l.w);
            fx_track_h(amax, h.x, true); fx_track_h(amax, h.y, true); fx_track_h(amax, h.z, true); fx_track_h(amax, h.w, true);      // (on the high parts: bx_split.hpp)
            if (it_rc[i] >= 0) {
                unsigned char* p = smem_bx + it_lds[i];
                *reinterpret_cast<uint4*>(p) = h;
                *reinterpret_cast<uint4*>(p + SPLB) = l;
            }
        }
    };

    int vid = blockIdx.x;
    if (vid >= total) return;
    issue_loads(vid);
    for (;;) {
        int b, iy0, ix0;
        tile_of(vid, b, iy0, ix0);
        stage_write();
        __syncthreads();
        const int nvid = vid + (int)gridDim.x;
        if (nvid < total) issue_loads(nvid);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        struct Frag { frag_t x[NXS]; frag_t wl; };
        Frag f[2];
        auto load = [&](int s, Frag& o) {
            const int k0 = Cfg::koff(2 * s), dk = Cfg::koff(2 * s + 1) - k0;
            const unsigned char* p = smem_bx + (lane_off + half * dk) + k0;
            o.wl = *reinterpret_cast<const frag_t*>(wl_lane + s * 1024);
#pragma unroll
            for (int q = 0; q < NXS; ++q) o.x[q] = *reinterpret_cast<const frag_t*>(p + q * SPLB);
        };
        load(0, f[0]);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const Frag& c = f[s & 1];
            if (s + 1 < NSTEP) load(s + 1, f[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            acc = mfma(c.wl, c.x[0], acc);          // small terms first: (2^11 w - q0) xh, w xl, q0 xh
            acc = mfma(wf[s][1], c.x[1], acc);
            acc = mfma(wf[s][0], c.x[0], acc);
            __builtin_amdgcn_sched_barrier(0);
        }
        // idle slots before the epilogue's address arithmetic: it must not land in operand registers of the last MFMAs
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc));      // (tied to the accumulator: see conv_bx_kernel)
        __builtin_amdgcn_sched_barrier(0);
        // ---- bias, ReLU, buffer stores: lane (pixel, half) holds couts 32 cb + (r & 3) + 8 (r >> 2) + 4 half ----------------------------
        {
            float bs[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 t = *reinterpret_cast<const float4*>(bias_lds + cb * 32 + 8 * g4 + 4 * half);
                bs[4 * g4] = t.x; bs[4 * g4 + 1] = t.y; bs[4 * g4 + 2] = t.z; bs[4 * g4 + 3] = t.w;
            }
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)b * COUT * HWo), 0, (int)(COUT * HWo * sizeof(float)), 0x00020000);
            const int oy = (iy0 >> 1) + orow, ox = (ix0 >> 1) + ocol;
            const int voff = oy < a.Ho && ox < a.Wo ? (int)(((size_t)(4 * half) * HWo + (size_t)oy * a.Wo + ox) * 4) : (int)0x80000000;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float y = fmaf(acc[r], FX_SCALE_INV, bs[r]);
                if (a.relu) y = fmaxf(y, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rs_out, voff, (int)((cb * 32 + (r & 3) + 8 * (r >> 2)) * HWo * 4), 0);
            }
        }
        if (nvid >= total) break;
        __syncthreads();
        vid = nvid;
    }
    fx_report_h(amax, a.status);
}

template <int CIN, bool IN_CL>
static int run_bxs2(const ConvW& c, const float* in, int B, int H, int W, float* out, hipStream_t st, int* status) {
    using Cfg = BxS2Cfg<CIN>;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if ((size_t)CIN * H * W * sizeof(float) >= 0x7fffffffu || (size_t)64 * Ho * Wo * sizeof(float) >= 0x7fffffffu) return -1;
    BxS2Args a;
    a.cold = g_debug_cold;
    a.status = status;
    a.in = in; a.wfrag = reinterpret_cast<const uint4*>(c.w_fx); a.bias = c.bias; a.out = out; a.relu = c.relu;
    a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.B = B;
    a.tiles_x = ceil_div(W, 32);
    a.tiles = a.tiles_x * ceil_div(H, 8);
    static AttrMask attr_done = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(conv_bxs2_kernel<CIN, IN_CL>), Cfg::LDS_BYTES, attr_done);
    const int total = xcd_grid_size(a.tiles, B);
    int grid = 2 * num_cus();
    if (grid > total) grid = total;
    conv_bxs2_kernel<CIN, IN_CL><<<grid, 256, Cfg::LDS_BYTES, st>>>(a);
    return 0;
}

template <int CIN, int COUT>
static int run_bx(const ConvW& c, const float* in, int B, int H, int W, float* out, hipStream_t st, long long* trace, int* status) {
    using Cfg = BxCfg<CIN, COUT>;
    if ((size_t)CIN * H * W * sizeof(float) >= 0x7fffffffu) return -1;      // buffer-resource range
    BxArgs a;
    a.cold = g_debug_cold;
    a.status = status;
    a.in = in; a.wfrag = reinterpret_cast<const uint4*>(c.w_fx); a.bias = c.bias; a.out = out; a.relu = c.relu; a.H = H; a.W = W; a.B = B; a.trace = trace;
    a.lag = 11;
    a.tiles_x = ceil_div(W, Cfg::TW);
    a.tiles = a.tiles_x * ceil_div(H, Cfg::TH);
    static AttrMask attr_done = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(conv_bx_kernel<CIN, COUT>), Cfg::LDS_BYTES, attr_done);
    const int total = xcd_grid_size(a.tiles, B);
    int grid = 2 * num_cus();            // two resident workgroups per CU; a multiple of 8 keeps a workgroup on its XCD
    if (grid > total) grid = total;
    conv_bx_kernel<CIN, COUT><<<grid, 256, Cfg::LDS_BYTES, st>>>(a);
    return 0;
}

template <int CIN, int COUT, bool IN_CL, bool OUT_CL>
static int run_bxd(const ConvW& c, const float* in, int B, int H, int W, float* out, hipStream_t st, int* status) {
    using Cfg = BxdCfg<CIN, COUT>;
    if ((size_t)CIN * H * W * sizeof(float) >= 0x7fffffffu) return -1;      // buffer-resource range
    BxArgs a;
    a.cold = g_debug_cold;
    a.status = status;
    a.in = in; a.wfrag = reinterpret_cast<const uint4*>(c.w_fx); a.bias = c.bias; a.out = out; a.relu = c.relu; a.H = H; a.W = W; a.B = B; a.trace = nullptr;
    a.lag = 0;
    a.tiles_x = ceil_div(W, Cfg::TW);
    a.tiles = a.tiles_x * ceil_div(H, Cfg::TH);
    static AttrMask attr_done = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(conv_bxd_kernel<CIN, COUT, IN_CL, OUT_CL>), Cfg::LDS_BYTES, attr_done);      // (a mask per instantiation)
    const int total = xcd_grid_size(a.tiles, B);
    int grid = num_cus();                // one workgroup (eight waves) per CU; a multiple of 8 keeps a workgroup on its XCD
    if (grid > total) grid = total;
    conv_bxd_kernel<CIN, COUT, IN_CL, OUT_CL><<<grid, 512, Cfg::LDS_BYTES, st>>>(a);
    return 0;
}

int bx_steps(int cin) { return (9 * (cin / 8) + 1) / 2; }

// -1: not one of this file's layers, or the layer has no fp16-pair weights (a |w| >= kFxMaxWeight): the caller falls back to the f32-MFMA kernel
// in_cl / out_cl: the input / output is channels-last (the backbone's links between these layers: conv_bx_links); the stamped round-4 kernel knows planes only
bool conv_bx_links(const ConvW& c, bool tracing) { return !tracing && c.ks == 3 && c.w_fx && c.cin == 24 && ((c.stride == 1 && c.cout == 24) || (c.stride == 2 && c.cout == 64)); }
int launch_conv_bx(const ConvW& c, const float* in, int B, int H, int W, float* out, hipStream_t st, long long* trace, int* status, bool in_cl, bool out_cl) {
    if (c.ks != 3 || !c.w_fx) return -1;
    if (c.stride == 1 && c.cin == 24 && c.cout == 24) {
        if (trace) return in_cl || out_cl ? -1 : run_bx<24, 24>(c, in, B, H, W, out, st, trace, status);      // (the stamped kernel is the round-4 form)
        if (in_cl) return out_cl ? run_bxd<24, 24, true, true>(c, in, B, H, W, out, st, status) : -1;      // (channels-last in, planes out: no layer of the network)
        return out_cl ? run_bxd<24, 24, false, true>(c, in, B, H, W, out, st, status) : run_bxd<24, 24, false, false>(c, in, B, H, W, out, st, status);
    }
    if (c.stride == 2 && c.cin == 24 && c.cout == 64 && !out_cl) return in_cl ? run_bxs2<24, true>(c, in, B, H, W, out, st, status) : run_bxs2<24, false>(c, in, B, H, W, out, st, status);
    return -1;
}

}  // namespace xfh
