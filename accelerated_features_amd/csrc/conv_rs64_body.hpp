// conv_rs64_kernel (k_conv_rs64.hip): 3x3 / stride 1 / 64 -> 64 convolution (block4.1, block4.2, block_fusion.0; modules/model.py:69-70,84) in the fp16-pair
// arithmetic (bx_split.hpp) with the WEIGHTS RESIDENT IN REGISTERS.  The body sits in a header so that tests/emu/ compiles the same source for the host.
//
// Why another kernel for these layers: conv_bx64_kernel streams all 216 KiB of split weights through LDS for EVERY 128-pixel unit (twelve tap rows, each behind a
// barrier and a DMA wait), which is what its small-map launches pay for (1/16 scale: 0.16 of the matrix floor) and a good part of the large ones (0.30).  Here a
// workgroup is four waves, one per SIMD (512 registers each), and the K = 576 of the product is split FOUR ways:
//   * wave w holds the weights of input channels 16 w .. 16 w + 15 -- all 9 taps x 64 couts x 3 fragments = 54 A operands of v_mfma_f32_32x32x16_f16, 216 registers,
//     loaded once per workgroup -- and multiplies only its own channel chunk: per 32 pixels and tap 2 ds_read_b128 (high, low parts) feed 6 MFMAs;
//   * so a wave also STAGES only its own chunk, into a ring of its own: no barrier between staging and use (LDS operations of a wave are ordered);
//   * the map is walked in PADDED RASTER order (pitch P = W + 2: one zero column left and right): the input of output position p under tap (dy, dx) is the staged
//     position p + dy P + dx -- one constant shift per tap, no tile halo, every input pixel is split once per run (+ 2 P + 2 at its start).  The two pad positions
//     of a row are computed and dropped (2 / P of the work);
//   * the four partial sums of a 32-position x 64-cout block meet in LDS: wave w owns couts 16 w .. + 15 (8 of its accumulator registers), writes the other 24
//     to their owners' slots (6 ds_write_b128), and after the block's ONE barrier adds three partials to its own (double-buffered: 48 KiB).
// Per block and wave: 54 MFMAs (1728 matrix-pipe cycles), 18 + 6 LDS reads, 6 LDS writes.  LDS: 48 KiB + 4 x nseg x 5 KiB of rings (nseg = 2 + (2 P + 1) / 64
// segments of 64 positions = exactly one unit's window: the next unit's new segment is converted inside the unit's MFMAs and stored behind its last operand read, over the
// segment the unit started with): 128 KiB at P = 82 (VGA 1/8 scale) -- one workgroup per CU; maps wider than 125 columns (nseg > 5) stay with conv_bx64_kernel.
//
// How a unit (64 positions = two blocks of 32) is laid out in time -- ONE basic block of 108 MFMAs, everything else placed between them (one wave per SIMD: nobody else hides a gap):
//   first block:   taps 0-2 | partial sums of the block BEFORE -> LDS | taps 3-4 | LDS-only barrier, read three partials | taps 5-8 + sum, bias, ReLU, stores of the block before
//   second block:  taps 0-2 + conversion of the next unit's new segment | partial sums of the first block -> LDS | taps 3-4 | barrier, read | taps 5-8 + finish of the first block
//                  + loads of the segment after next | the converted segment -> ring (over the segment this unit started with)
// A tap's B operands are read TWO taps ahead (taps 7 / 8 read the next block's taps 0 / 1), straight into accumulation registers (XFH_AGPR): registers no vector-ALU result is allocated
// to, so no idle slots guard the matrix core's operand reads.  The four waves run four COPIES of this code (template parameter wave): which accumulator registers go to whom is static.
// Forms: FUSE 1 / 2 = the trailing 1x1 (block3.2 NCHW, block_fusion.2 channels-last) on v_mfma_f32_16x16x32_f16, a block's 3x3 outputs handed over as fp16 pairs through LDS
// (Y buffers), its 1x1 two blocks later; CIN 128 = block5.1 / block5.2: a workgroup per cout QUARTER, a wave multiplies 32 channels (two chunks, an accumulator each);
// TRACE = the stamped twin (xfh_debug_trace).  tests/test_conv_rs64_emulated.py, tools/fuzz_conv_rs64_emulated.py, tools/bench_src/rs64_probe.cpp.
#pragma once
#ifndef XFH_HOST_EMU
#include "kernels.hpp"
#ifndef XFH_DYN_LDS_BYTES
#define XFH_DYN_LDS_BYTES(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif
/* a wave's LDS operations are ordered and its lanes run in lock-step: a position one lane wrote is there when another lane reads it -- nothing to wait for (the host
   emulation, where lanes are threads, meets here) */
#define XFH_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
/* workgroup barrier for LDS traffic alone: the wave's global loads and stores stay in flight (__syncthreads waits for them too) */
#define XFH_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define XFH_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
/* a value whose home is an accumulation register: the matrix instructions read their A operand from there directly; left to itself hipcc parks the weights that do not fit into
   the 256 architectural registers there too, but copies each fragment back (four v_accvgpr_read) in front of every use */
#define XFH_AGPR(x) asm volatile("" : "+a"(x))
#define XFH_VGPR(x) asm volatile("" : "+v"(x))
/* the end of a tap: the NEXT tap's B operands (read while this tap was multiplied) and the accumulators in one statement -- it follows this tap's MFMAs (it takes their results)
   and precedes the next tap's, so the wait for the LDS read sits BEHIND the six MFMAs it travelled under (a pin right behind the read puts the wait in front of them: the matrix
   pipe then idles one LDS latency per tap) */
#define XFH_AGPR_TAP(h, l, c0, c1) asm volatile("" : "+a"(h), "+a"(l), "+a"(c0), "+a"(c1))
/* the start of a tap, behind the read of the next tap's operands: the tap's MFMAs take the accumulators from here, so they are issued behind the read (left to itself the
   instruction selector lists the six MFMAs first and the read right in front of its wait) */
#define XFH_AGPR_ACC(c0, c1) asm volatile("" : "+a"(c0), "+a"(c1))
#define XFH_AGPR_TAP2(h, l, h1, l1, c0, c1) asm volatile("" : "+a"(h), "+a"(l), "+a"(h1), "+a"(l1), "+a"(c0), "+a"(c1))
#endif
#include <type_traits>
#include "bx_split.hpp"

namespace xfh {

struct Rs64Args {
    const float* in;
    const void* wq;            // [wave = channel chunk 4][tap 9][cout block 2][fragment 3][64 lanes][8 fp16]   (weight_split.hpp: pack_rs64)
    const float* bias;
    float* out;
    int relu, H, W, B;
    int ns, ws;                // column STRIPS per image and their width: a map wider than the rings reach (125 columns; 93 with the fused 1x1) runs as ns strips of ws columns,
                               // each a padded raster of its own whose pad columns hold the neighbouring strips' pixels (zeros only beyond the map); ns = 1, ws = W otherwise
    int P;                     // ws + 2
    float inv_p;               // 1 / P
    int nu;                    // 64-position units per image: ceil(H P / 64)
    int nseg;                  // ring capacity in segments
    int k;                     // runs per image (a run = consecutive units of one image, one workgroup)
    int cold;
    int* status;               // range guard of the fp16 pair (bx_split.hpp), may be NULL
    // fused trailing 1x1 (64 -> 64; FUSE 1: NCHW output, 2: channels-last): [wave = 16 couts][K step 2][fragment 3][64 lanes][8 fp16] (weight_split.hpp: pack_rs64_1x1)
    const void* wq2;
    const float* bias2;
    int relu2;
    long long* trace;          // debug (the TRACE instantiations only): s_memtime stamps of wave 0, 32 per workgroup (see conv_rs64_wave)
};

namespace rs64 {
// staged position: a wave's channels (16; 32 for the 128-channel layers) x (high, low) fp16 + 16: an odd multiple of 16 B (distinct banks for the 16 lanes of a ds_read_b128 group)
template <int CIN> constexpr int pixb() { return CIN == 128 ? 144 : 80; }
constexpr int PIXB = 80;
constexpr int SEG_PX = 64, SEG_BYTES = SEG_PX * PIXB;         // 5120
template <int CIN> constexpr int seg_bytes() { return SEG_PX * pixb<CIN>(); }                                  // 5120 | 9216
constexpr int RED_BYTES = 4 * 3 * 2048;                       // [owner 4][source slot 3][part 2][64 lanes] float4
constexpr int Y_PITCH = 272, Y_BYTES = 32 * Y_PITCH;           // fused 1x1: a block's 3x3 outputs as fp16 pairs, [position 32][high parts of the 64 channels | low parts] + 16 (bank spread)
template <int FUSE> constexpr int ring_off() { return 2 * RED_BYTES + (FUSE ? 2 * Y_BYTES : 0); }      // 49152 | 66560
constexpr int MAX_NSEG = 5;
// behind a wave's ring: a MIRROR of its first 34 positions, so that the 32 positions of a tap (and the + 1, + 2 of the taps to its right, which share its address register)
// never straddle the ring's end -- the wrap is a wave-uniform decision (scalar ALU), one address computation serves a tap ROW
constexpr int MIRROR_PX = 34;
template <int CIN> constexpr int mirror_bytes() { return MIRROR_PX * pixb<CIN>(); }                           // 2720 | 4896
template <int CIN> constexpr int ring_stride(int nseg) { return nseg * seg_bytes<CIN>() + mirror_bytes<CIN>(); }
constexpr int WQ_HALFS = 4 * 9 * 2 * 3 * 64 * 8;              // 110592 fp16 = 216 KiB
inline int nseg_for(int P) { return 2 + (2 * P + 1) / 64; }
// column strips of a map of width W for rings of at most max_seg segments: the fewest strips whose width fits (equal widths, the last one takes what is left)
inline void strips_for(int W, int max_seg, int& ns, int& ws) {
    for (ns = 1;; ++ns) {
        ws = (W + ns - 1) / ns;
        if (nseg_for(ws + 2) <= max_seg) return;
    }
}
inline int lds_bytes(int nseg, bool fuse = false) { return (fuse ? ring_off<1>() : ring_off<0>()) + 4 * ring_stride<64>(nseg); }
inline int max_nseg(bool fuse) { return fuse ? 4 : 5; }
// CIN = COUT = 128 (block5.1, block5.2): a workgroup computes a QUARTER of the couts (32) and a wave multiplies 32 input channels (two 16-channel chunks, one accumulator each)
constexpr int RED128_BYTES = 4 * 3 * 1024;                    // [owner 4][source slot 3][64 lanes] float4: wave w owns couts 8 w .. + 7 of the quarter (4 registers per lane)
inline int lds_bytes128(int nseg) { return 2 * RED128_BYTES + 4 * ring_stride<128>(nseg); }
constexpr int MAX_NSEG128 = 3;
static_assert(2 * RED128_BYTES + 4 * ring_stride<128>(MAX_NSEG128) <= 160 * 1024, "LDS of a CU");
// runs per image for `grid` workgroups: whole images while there are enough of them, else every image in grid / B parts (at least one unit each)
inline int runs_per_image(int B, int nu, int grid) { const int k = B >= grid ? 1 : grid / B; return k < 1 ? 1 : k > nu ? nu : k; }
static_assert(ring_off<0>() + 4 * ring_stride<64>(5) <= 160 * 1024 && ring_off<1>() + 4 * ring_stride<64>(4) <= 160 * 1024, "LDS of a CU");
}

// the code of ONE wave of the workgroup (wave = its K quarter and the couts it finishes): four copies, so that which accumulator registers are a wave's own and which go to
// whom is static (selected at run time it costs a v_cndmask per register and use, or a branch tree in the middle of the MFMA stream)
// TRACE (a separate instantiation: the stamps' branches would cut the unit's basic block in the production code): lane 0 of wave 0 writes s_memtime to trace[128 workgroup + k] (k = 32 + 27 block + slot: the slots of the second unit):
// k = 0 entry, 1 weights in registers, 2 the first run's ring filled; of the workgroup's SECOND unit: 3 start, 4 / 5 / 6 / 7 first block (taps 0-2 issued, taps 3-4 issued =
// at the barrier, barrier passed, taps 5-8 + reduction issued), 8 .. 11 the same of the second block, 12 unit end (segment stored); 13 exit, 14 = units this workgroup processed
template <int wave, int FUSE, int CIN = 64, bool TRACE = false>
__device__ __forceinline__ void conv_rs64_wave(const Rs64Args& a) {
    using namespace rs64;
    constexpr bool C128 = CIN == 128;            // 128 -> 128: the workgroup's cout quarter a.cq-th of four is part of the run index; this wave multiplies channels 32 wave .. + 31
    static_assert(!C128 || FUSE == 0, "the 128-channel form has no fused 1x1 (a workgroup holds a quarter of the 3x3's outputs)");
    constexpr int PIXB = pixb<CIN>(), SEG_BYTES = seg_bytes<CIN>(), RED_BYTES = C128 ? RED128_BYTES : rs64::RED_BYTES;
    constexpr int LO_OFF = C128 ? 64 : 32;       // low parts behind the high parts of the wave's channels
    constexpr int RING_OFF = C128 ? 2 * RED128_BYTES : ring_off<FUSE>();
    XFH_DYN_LDS_BYTES(smem_rs);
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, kg = lane >> 5;
    const int P = a.P, H = a.H, W = a.W, HW = H * W;
    const float inv_p = a.inv_p;
    long long* tr = TRACE && wave == 0 && a.trace && lane == 0 ? a.trace + (size_t)blockIdx.x * 128 : nullptr;
    int n_units = 0;
#define RS_STAMP(k) { if constexpr (TRACE) { if (tr) tr[k] = __builtin_amdgcn_s_memtime(); } }
#define RS_STAMP_U(k) { if constexpr (TRACE) { if (tr && n_units == 1) tr[k] = __builtin_amdgcn_s_memtime(); } }
    RS_STAMP(0)

    // ---- this wave's weights: channels 16 wave .. + 15 under every tap, all 64 couts, three fragments (q0, q1, q2 of split_weight mode 1)
    // (128 channels: the middle index is the 16-channel CHUNK of the wave's 32 channels instead of the cout block; the quarter's weights are selected per run -- every run of a
    // workgroup has the same quarter: the launcher makes the grid a multiple of four)
    f16x8 A[9][2][3];
    {
        const f16x8* wp = reinterpret_cast<const f16x8*>(a.wq) + ((size_t)(C128 ? (int)(blockIdx.x & 3) * 4 : 0) + wave) * (9 * 2 * 3 * 64) + lane;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 3; ++q) A[t][cb][q] = wp[((t * 2 + cb) * 3 + q) * 64];
    }
    RS_STAMP(1)
    // ---- the couts this wave finishes: 16 wave + 8 (k >> 2) + 4 kg + (k & 3), k = 0 .. 7 = registers 8 (wave & 1) + k of accumulator wave >> 1
    float bs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bs[k] = C128 ? a.bias[32 * (int)(blockIdx.x & 3) + 8 * wave + 4 * kg + (k & 3)] : a.bias[16 * wave + 8 * (k >> 2) + 4 * kg + (k & 3)];
    const float floor_y = a.relu ? 0.f : -__builtin_inff();

    unsigned char* ring = smem_rs + RING_OFF + wave * ring_stride<CIN>(a.nseg);
    const unsigned Rb = (unsigned)a.nseg * SEG_BYTES;
    const unsigned lane_b = (unsigned)(n * PIXB + kg * 16);
    unsigned amax = 0;
    float amaxf = 0.f;                             // range guard of the staged segments (on the fp32 values: one v_max3_f32 per pair)
    int kb = 0;                                    // blocks this workgroup has reduced: parity = reduction buffer
    const int nruns = a.B * a.ns * a.k;

    // position i of a padded raster = (row, column): i < 2^20, P >= 3: (i + 0.5) / P is at least 1 / (2 P) away from an integer, the product's error is below 1e-4
    auto row_of = [&](int i) { return (int)(((float)i + 0.5f) * inv_p); };
    // the B operands of a tap: positions t0 + shift + n of the ring, high parts and low parts of this lane's 8 channels
    struct Xf { f16x8 h, l, h1, l1; };            // (h1, l1: the second channel chunk of the 128-channel form)
    // the B operands of a tap ROW: rowb = byte offset of this lane's position under the row's LEFT tap; the taps to its right read PIXB, 2 PIXB further on (immediate offsets)
    auto row_base = [&](unsigned tb /* byte offset of the block's first position under the row's left tap, < 2 Rb */) __attribute__((always_inline)) {
        tb = tb >= Rb ? tb - Rb : tb;                                                 // (wave-uniform; the positions behind the ring's end are in its mirror)
        return tb + lane_b;
    };
    auto ldx = [&](unsigned rowb, auto DXC, Xf& x, bool pin = true) __attribute__((always_inline)) {
        constexpr int dxo = decltype(DXC)::value * PIXB;
        x.h = *reinterpret_cast<const f16x8*>(ring + rowb + dxo);
        x.l = *reinterpret_cast<const f16x8*>(ring + rowb + dxo + LO_OFF);
        if constexpr (C128) {
            x.h1 = *reinterpret_cast<const f16x8*>(ring + rowb + dxo + 32);
            x.l1 = *reinterpret_cast<const f16x8*>(ring + rowb + dxo + LO_OFF + 32);
            if (pin) { XFH_AGPR(x.h1); XFH_AGPR(x.l1); }
        }
        // the B operands live in accumulation registers too (ds_read writes them there directly): registers no vector-ALU result is ever allocated to, so none can land in
        // an operand the matrix core is still reading (tools/check_mfma_war.py) -- without idle slots or keep-alive fences in the MFMA stream
        if (pin) { XFH_AGPR(x.h); XFH_AGPR(x.l); }
    };
    unsigned rowshift_b[3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) rowshift_b[dy] = (unsigned)(dy * P * PIXB);
    unsigned rowb = 0;                                   // the current tap row's base (a register that lives across three taps -- and across the block boundary for row 0)

    // where a finished block goes: the block's accumulators wait one block long for their reduction (it runs inside the next block's MFMAs)
    struct Pend { __amdgpu_buffer_rsrc_t rs; int voff; };
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

    // ---- the partial sums of a block meet: registers 8 (o & 1) .. + 7 of accumulator o >> 1 belong to wave o.  red_write: the 24 foreign ones to their owners' slots
    // (buffer kb & 1); red_read: the three partials for this wave's couts; red_finish: sum, bias, ReLU, stores
    auto red_write = [&](const f32x16& c0, const f32x16& c1) __attribute__((always_inline)) {
        unsigned char* red = smem_rs + (kb & 1) * RED_BYTES;
        if constexpr (C128) {      // one cout block, two chunk accumulators: registers 4 o .. + 3 of their sum belong to wave o
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                if (o == wave) continue;
                float4* d = reinterpret_cast<float4*>(red + o * (3 * 1024) + (wave < o ? wave : wave - 1) * 1024) + lane;
                d[0] = make_float4(c0[4 * o] + c1[4 * o], c0[4 * o + 1] + c1[4 * o + 1], c0[4 * o + 2] + c1[4 * o + 2], c0[4 * o + 3] + c1[4 * o + 3]);
            }
            return;
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            if (o == wave) continue;
            float4* d = reinterpret_cast<float4*>(red + o * (3 * 2048) + (wave < o ? wave : wave - 1) * 2048) + lane;      // this wave is source slot wave - (wave > o) of owner o
            const f32x16& c = (o >> 1) ? c1 : c0;
            d[0] = make_float4(c[8 * (o & 1)], c[8 * (o & 1) + 1], c[8 * (o & 1) + 2], c[8 * (o & 1) + 3]);
            d[64] = make_float4(c[8 * (o & 1) + 4], c[8 * (o & 1) + 5], c[8 * (o & 1) + 6], c[8 * (o & 1) + 7]);
        }
    };
    auto red_read = [&](float4 (&part)[3][2]) __attribute__((always_inline)) {
        const unsigned char* red = smem_rs + (kb & 1) * RED_BYTES;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if constexpr (C128) { part[s][0] = reinterpret_cast<const float4*>(red + wave * (3 * 1024) + s * 1024)[lane]; continue; }
            const float4* d = reinterpret_cast<const float4*>(red + wave * (3 * 2048) + s * 2048) + lane;
            part[s][0] = d[0]; part[s][1] = d[64];
        }
    };
    auto red_finish = [&](const f32x16& c0, const f32x16& c1, const float4 (&part)[3][2], const Pend& pd, int ybuf) __attribute__((always_inline)) {
        if constexpr (C128) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float p3[3] = {k == 0 ? part[0][0].x : k == 1 ? part[0][0].y : k == 2 ? part[0][0].z : part[0][0].w, k == 0 ? part[1][0].x : k == 1 ? part[1][0].y : k == 2 ? part[1][0].z : part[1][0].w,
                                     k == 0 ? part[2][0].x : k == 1 ? part[2][0].y : k == 2 ? part[2][0].z : part[2][0].w};
                const float sum = (((c0[4 * wave + k] + c1[4 * wave + k]) + p3[0]) + p3[1]) + p3[2];
                const float y = fmaxf(sum * FX_SCALE_INV + bs[k], floor_y);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), pd.rs, pd.voff, k * HW * 4, 0);
            }
            return;
        }
        float own[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) own[k] = ((wave >> 1) ? c1 : c0)[8 * (wave & 1) + k];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            own[0] += part[s][0].x; own[1] += part[s][0].y; own[2] += part[s][0].z; own[3] += part[s][0].w;
            own[4] += part[s][1].x; own[5] += part[s][1].y; own[6] += part[s][1].z; own[7] += part[s][1].w;
        }
        float y[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = fmaxf(own[k] * FX_SCALE_INV + bs[k], floor_y);
        if constexpr (FUSE == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y[k]), pd.rs, pd.voff, ((k & 3) + 8 * (k >> 2)) * HW * 4, 0);
        } else {
            // the 1x1's B operands: this lane's channels 16 wave + 4 kg + {0 .. 3} and 16 wave + 8 + 4 kg + {0 .. 3} of position n, as fp16 pairs, into Y[ybuf]
            unsigned char* yp = smem_rs + 2 * RED_BYTES + ybuf * Y_BYTES + n * Y_PITCH + (16 * wave + 4 * kg) * 2;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                uint2 hh, ll;
                split2_f16_scalar(y[4 * g], y[4 * g + 1], hh.x, ll.x);
                split2_f16_scalar(y[4 * g + 2], y[4 * g + 3], hh.y, ll.y);
                fx_track_h(amax, hh.x, true); fx_track_h(amax, hh.y, true);
                *reinterpret_cast<uint2*>(yp + 16 * g) = hh;
                *reinterpret_cast<uint2*>(yp + 16 * g + 128) = ll;
            }
        }
    };
    // ---- fused 1x1 on a block whose 3x3 outputs wait in Y[ybuf] (written one block ago, published by the barrier since): this wave's 16 couts x 32 positions,
    // K = 64 as two steps of v_mfma_f32_16x16x32_f16 (lane: position l & 15 of the 16-position half, K values 8 (l >> 4) .. + 7; D: couts 4 (l >> 4) + j)
    struct Pend2 { __amdgpu_buffer_rsrc_t rs; int voff[2]; };
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f16x8 A2[2][3];
    float bs2[4];
    if constexpr (FUSE != 0) {
        const f16x8* wp2 = reinterpret_cast<const f16x8*>(a.wq2) + (size_t)wave * (2 * 3 * 64) + lane;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int q = 0; q < 3; ++q) A2[s2][q] = wp2[(s2 * 3 + q) * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) bs2[j] = a.bias2[16 * wave + 4 * (lane >> 4) + j];
    }
    const float floor_y2 = a.relu2 ? 0.f : -__builtin_inff();
    auto conv1x1 = [&](int ybuf, const Pend2& pd) __attribute__((always_inline)) {
        const unsigned char* yp = smem_rs + 2 * RED_BYTES + ybuf * Y_BYTES + (lane & 15) * Y_PITCH + (lane >> 4) * 16;
        // the two 16-position halves take turns (two accumulators): a dependent MFMA waits for its predecessor
        f32x4 d[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            f16x8 xh[2], xl[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                xh[nb] = *reinterpret_cast<const f16x8*>(yp + nb * 16 * Y_PITCH + s2 * 64);
                xl[nb] = *reinterpret_cast<const f16x8*>(yp + nb * 16 * Y_PITCH + s2 * 64 + 128);
                XFH_AGPR(xh[nb]); XFH_AGPR(xl[nb]);           // (as the 3x3's B operands: registers no vector-ALU result is allocated to)
            }
            d[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[s2][2], xh[0], d[0], 0, 0, 0);
            d[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[s2][2], xh[1], d[1], 0, 0, 0);
            d[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[s2][1], xl[0], d[0], 0, 0, 0);
            d[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[s2][1], xl[1], d[1], 0, 0, 0);
            d[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[s2][0], xh[0], d[0], 0, 0, 0);
            d[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[s2][0], xh[1], d[1], 0, 0, 0);
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            float z[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = fmaxf(d[nb][j] * FX_SCALE_INV + bs2[j], floor_y2);
            if constexpr (FUSE == 2) {
                typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
                const u32x4s q = {__float_as_uint(z[0]), __float_as_uint(z[1]), __float_as_uint(z[2]), __float_as_uint(z[3])};
                __builtin_amdgcn_raw_buffer_store_b128(q, pd.rs, pd.voff[nb], 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(z[j]), pd.rs, pd.voff[nb], j * HW * 4, 0);
            }
        }
    };
    // ---- the fused 1x1 in PIECES for the slots of the 3x3's MFMA stream (below): slots 16 / 17 read the two K halves of Y[ybuf], 18 .. 23 one MFMA pair each
    // (both 16-position halves: two accumulators), 24 / 25 finish and store one half each.  Its twelve small MFMAs join the stream (16 matrix-pipe cycles each); run
    // behind the block instead, reads -> wait -> MFMAs -> stores cost 470 cycles per block with the pipe mostly idle (87 us against 68 for the 3x3 alone)
    struct C1x1 { f16x8 xh[2][2], xl[2][2]; f32x4 d[2]; };
    auto conv1x1_piece = [&](auto IDC, C1x1& w, int ybuf, const Pend2& pd) __attribute__((always_inline)) {
        constexpr int id = decltype(IDC)::value;
        const unsigned char* yp = smem_rs + 2 * RED_BYTES + ybuf * Y_BYTES + (lane & 15) * Y_PITCH + (lane >> 4) * 16;
        if constexpr (id == 16 || id == 17) {
            constexpr int s2 = id - 16;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                w.xh[s2][nb] = *reinterpret_cast<const f16x8*>(yp + nb * 16 * Y_PITCH + s2 * 64);
                w.xl[s2][nb] = *reinterpret_cast<const f16x8*>(yp + nb * 16 * Y_PITCH + s2 * 64 + 128);
            }
        }
        if constexpr (id >= 18 && id < 24) {
            constexpr int s2 = (id - 18) / 3, q = 2 - (id - 18) % 3;        // fragment order q2 (high parts), q1 (low parts), q0 (high parts) as in conv1x1
            if constexpr (q == 2) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) { XFH_AGPR(w.xh[s2][nb]); XFH_AGPR(w.xl[s2][nb]); }
            }
            if constexpr (id == 18) {
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                w.d[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[0][2], w.xh[0][0], zero, 0, 0, 0);
                w.d[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[0][2], w.xh[0][1], zero, 0, 0, 0);
            } else {
                XFH_AGPR(w.d[0]); XFH_AGPR(w.d[1]);
                w.d[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[s2][q], q == 1 ? w.xl[s2][0] : w.xh[s2][0], w.d[0], 0, 0, 0);
                w.d[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A2[s2][q], q == 1 ? w.xl[s2][1] : w.xh[s2][1], w.d[1], 0, 0, 0);
            }
            XFH_AGPR(w.d[0]); XFH_AGPR(w.d[1]);
        }
        if constexpr (id == 24 || id == 25) {
            constexpr int nb = id - 24;
            float z[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = fmaxf(w.d[nb][j] * FX_SCALE_INV + bs2[j], floor_y2);
            if constexpr (FUSE == 2) {
                typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
                const u32x4s qv = {__float_as_uint(z[0]), __float_as_uint(z[1]), __float_as_uint(z[2]), __float_as_uint(z[3])};
                __builtin_amdgcn_raw_buffer_store_b128(qv, pd.rs, pd.voff[nb], 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(z[j]), pd.rs, pd.voff[nb], j * HW * 4, 0);
            }
        }
    };
    // ---- the reduction in PIECES, one per slot of the MFMA stream (below): piece p of a block's foreign partial sums (6 ds_write_b128; 128 channels: 3) ...
    constexpr int NPIECE = C128 ? 3 : 6, NSTORE = C128 ? 4 : 8;
    auto red_write_piece = [&](auto PC, const f32x16& c0, const f32x16& c1) __attribute__((always_inline)) {
        constexpr int p = decltype(PC)::value;
        unsigned char* red = smem_rs + (kb & 1) * RED_BYTES;
        if constexpr (C128) {
            constexpr int o = p < wave ? p : p + 1;
            float4* d = reinterpret_cast<float4*>(red + o * (3 * 1024) + (wave < o ? wave : wave - 1) * 1024) + lane;
            d[0] = make_float4(c0[4 * o] + c1[4 * o], c0[4 * o + 1] + c1[4 * o + 1], c0[4 * o + 2] + c1[4 * o + 2], c0[4 * o + 3] + c1[4 * o + 3]);
        } else {
            constexpr int q = p >> 1, o = q < wave ? q : q + 1, half = p & 1;
            float4* d = reinterpret_cast<float4*>(red + o * (3 * 2048) + (wave < o ? wave : wave - 1) * 2048) + lane;
            const f32x16& c = (o >> 1) ? c1 : c0;
            d[64 * half] = make_float4(c[8 * (o & 1) + 4 * half], c[8 * (o & 1) + 4 * half + 1], c[8 * (o & 1) + 4 * half + 2], c[8 * (o & 1) + 4 * half + 3]);
        }
    };
    // ... the finished values of this wave's couts (sum of the four partials, bias, ReLU) and their stores one by one (the unfused forms)
    auto finish_values = [&](const f32x16& c0, const f32x16& c1, const float4 (&part)[3][2], float (&y)[8]) __attribute__((always_inline)) {
        if constexpr (C128) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float p3[3] = {k == 0 ? part[0][0].x : k == 1 ? part[0][0].y : k == 2 ? part[0][0].z : part[0][0].w, k == 0 ? part[1][0].x : k == 1 ? part[1][0].y : k == 2 ? part[1][0].z : part[1][0].w,
                                     k == 0 ? part[2][0].x : k == 1 ? part[2][0].y : k == 2 ? part[2][0].z : part[2][0].w};
                const float sum = (((c0[4 * wave + k] + c1[4 * wave + k]) + p3[0]) + p3[1]) + p3[2];
                y[k] = fmaxf(sum * FX_SCALE_INV + bs[k], floor_y);
            }
        } else {
            float own[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) own[k] = ((wave >> 1) ? c1 : c0)[8 * (wave & 1) + k];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                own[0] += part[s][0].x; own[1] += part[s][0].y; own[2] += part[s][0].z; own[3] += part[s][0].w;
                own[4] += part[s][1].x; own[5] += part[s][1].y; own[6] += part[s][1].z; own[7] += part[s][1].w;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) y[k] = fmaxf(own[k] * FX_SCALE_INV + bs[k], floor_y);
        }
    };
    auto finish_store = [&](auto KC, const float (&y)[8], const Pend& pd) __attribute__((always_inline)) {
        constexpr int k = decltype(KC)::value;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y[k]), pd.rs, pd.voff, (C128 ? k : (k & 3) + 8 * (k >> 2)) * HW * 4, 0);
    };
    // ---- ONE tap of a block = six MFMAs in three pairs, a SLOT behind each pair (27 per block).  The operands of tap t + 1 are requested first and awaited last (x[t & 1] <->
    // x[(t + 1) & 1]; behind tap 8: tap 0 of the NEXT block -- 32 positions on: the second block of the unit, or the first of the next unit, whose segment has been in the ring
    // since this unit began).  The pins (XFH_AGPR_ACC / _TAP) take and return the accumulators, so MFMA pairs and pins alternate in program order, and every MEMORY operation a
    // slot issues (LDS writes of the partial sums, global stores and loads, the ring's segment) stays between its two pairs: one wave per SIMD means nobody else fills the matrix
    // pipe while this wave transfers a burst of them (a ds_write_b128 is 13 cycles of the store path, a burst of six from each of the four waves 300: measured as + 270 cycles on
    // the two taps behind it; the eight stores and sixteen loads of a unit likewise).  Vector-ALU work is not ordered by the pins: the scheduler spreads it over the gaps.
    auto tap = [&](auto TC, auto PARC, unsigned t0b, f32x16& c0, f32x16& c1, Xf (&x)[3], auto&& slot) __attribute__((always_inline)) {
        // operands travel TWO taps ahead (three register sets; a block has nine taps, so every block starts on set 0): the read issued here is awaited at the end of the NEXT
        // tap, 384 matrix-pipe cycles on -- one tap ahead (192) the wait was still exposed whenever the read queued behind a burst of the partial sums' ds_write_b128
        // (PMC: the waves spent ~ 7 % of a unit parked at these waits)
        constexpr int t = decltype(TC)::value, cur = t % 3, nxt = (t + 1) % 3, far = (t + 2) % 3;
        (void)PARC;
        constexpr int ft = t + 2;                      // the tap whose operands leave LDS now: of this block (< 9), or tap ft - 9 (0 or 1) of the next one, 32 positions on
        if constexpr (ft % 3 == 0) rowb = row_base(ft < 9 ? t0b + rowshift_b[ft < 9 ? ft / 3 : 0] : t0b + 32 * PIXB);
        ldx(rowb, std::integral_constant<int, ft % 3>{}, x[far], false);
        if constexpr (t > 0) XFH_AGPR_ACC(c0, c1);      // (tap 0 starts from the literal zero: a pinned accumulator would have to be written first, 32 v_accvgpr_write per block)
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[t][0][2], x[cur].h, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[t][1][2], C128 ? x[cur].h1 : x[cur].h, c1, 0, 0, 0);
        XFH_AGPR_ACC(c0, c1);
        slot(std::integral_constant<int, 3 * t>{});
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[t][0][1], x[cur].l, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[t][1][1], C128 ? x[cur].l1 : x[cur].l, c1, 0, 0, 0);
        XFH_AGPR_ACC(c0, c1);
        slot(std::integral_constant<int, 3 * t + 1>{});
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[t][0][0], x[cur].h, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[t][1][0], C128 ? x[cur].h1 : x[cur].h, c1, 0, 0, 0);
        if constexpr (C128) { XFH_AGPR_TAP2(x[nxt].h, x[nxt].l, x[nxt].h1, x[nxt].l1, c0, c1); }
        else { XFH_AGPR_TAP(x[nxt].h, x[nxt].l, c0, c1); }
        slot(std::integral_constant<int, 3 * t + 2>{});
    };
    auto taps9 = [&](auto PARC, unsigned t0b, f32x16& c0, f32x16& c1, Xf (&x)[3], auto&& slot) __attribute__((always_inline)) {
        tap(std::integral_constant<int, 0>{}, PARC, t0b, c0, c1, x, slot); tap(std::integral_constant<int, 1>{}, PARC, t0b, c0, c1, x, slot); tap(std::integral_constant<int, 2>{}, PARC, t0b, c0, c1, x, slot);
        tap(std::integral_constant<int, 3>{}, PARC, t0b, c0, c1, x, slot); tap(std::integral_constant<int, 4>{}, PARC, t0b, c0, c1, x, slot); tap(std::integral_constant<int, 5>{}, PARC, t0b, c0, c1, x, slot);
        tap(std::integral_constant<int, 6>{}, PARC, t0b, c0, c1, x, slot); tap(std::integral_constant<int, 7>{}, PARC, t0b, c0, c1, x, slot); tap(std::integral_constant<int, 8>{}, PARC, t0b, c0, c1, x, slot);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I3 = std::integral_constant<int, 3>;
    using I5 = std::integral_constant<int, 5>; using I9 = std::integral_constant<int, 9>;

    f32x16 ca0, ca1, cb0, cb1;                     // accumulator sets of the unit's first / second block
#pragma unroll
    for (int i = 0; i < 16; ++i) { ca0[i] = 0.f; ca1[i] = 0.f; cb0[i] = 0.f; cb1[i] = 0.f; }
    Pend pend_a, pend_b;                            // (b: the second block of the unit BEFORE: nothing yet -- a resource of zero bytes drops the stores)
    pend_b.rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, 0, 0x00020000);
    pend_b.voff = (int)0x80000000;
    pend_a = pend_b;
    Pend2 p2a, p2b, p2a_prev, p2b_prev;             // fused 1x1: where the unit's first / second block goes, and the same of the unit before (a block's 1x1 runs a unit later)
    p2a.rs = pend_b.rs; p2a.voff[0] = p2a.voff[1] = (int)0x80000000;
    p2b = p2a; p2a_prev = p2a; p2b_prev = p2a;
    Xf x[3];

    constexpr int NV = C128 ? 32 : 16, NQ = NV / 8;      // values a lane stages per segment (its position's channels), 16-byte groups of their high / low parts
    const int cq = C128 ? (int)(blockIdx.x & 3) : 0;      // 128 channels: this workgroup's cout quarter (the grid is a multiple of four; the runs go to the groups of four)
    for (int run = C128 ? (int)(blockIdx.x >> 2) : (int)blockIdx.x; run < nruns; run += C128 ? (int)(gridDim.x >> 2) : (int)gridDim.x) {
        const int bs_i = run / a.k, part_i = run - bs_i * a.k;        // (image, strip), part of the strip's raster
        const int b = bs_i / a.ns, x0 = (bs_i - b * a.ns) * a.ws;     // the strip's first column
        const int wv = min(a.ws, W - x0);                             // its columns (the last strip of a map may be narrower)
        const int ua = (int)((long long)a.nu * part_i / a.k), ub = (int)((long long)a.nu * (part_i + 1) / a.k);
        if (ua >= ub) continue;
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + ((size_t)b * CIN + NV * wave) * HW), 0, (int)(NV * HW * sizeof(float)), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = C128 ? __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + ((size_t)b * 128 + 32 * cq + 8 * wave) * HW), 0, (int)(8 * HW * sizeof(float)), 0x00020000)
                                                   : __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + ((size_t)b * 64 + 16 * wave) * HW), 0, (int)(16 * HW * sizeof(float)), 0x00020000);
        // segment s of the image's padded raster: position 64 s + lane = (row r, column c) of the (H + 2) x P frame = pixel (r - 1, c - 1); outside the map: zeros
        auto seg_load = [&](int s, bool en, float (&v)[NV]) __attribute__((always_inline)) {
            const int i = 64 * s + lane, r = row_of(i), c = i - r * P;
            const int iy = r - 1, ix = x0 + c - 1;
            // (one select, no short-circuit: a branch here would cut the unit's basic block and strand the conversion and the loads behind the last MFMA)
            const bool inside = (int)en & (int)((unsigned)iy < (unsigned)H) & (int)((unsigned)ix < (unsigned)W);
            const int voff = inside ? (iy * W + ix) * 4 : (int)0x80000000;
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_in, voff, j * HW * 4, 0));
        };
        // a segment's values -> fp16 pairs (vector ALU work, placed inside the unit's MFMAs) ... and into the ring (four ds_write_b128, placed behind the unit's LAST operand read:
        // the segment replaces the one the unit itself started with -- the ring holds exactly one unit's window)
        auto seg_convert = [&](const float (&v)[NV], u32x4 (&h)[NQ], u32x4 (&l)[NQ]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NV / 2; ++j) {
                const float a0 = v[2 * j], a1 = v[2 * j + 1];
                unsigned hh, ll;
                split2_f16_scalar(a0, a1, hh, ll);
                fx_track(amaxf, a0, a1);
                h[j >> 2][j & 3] = hh; l[j >> 2][j & 3] = ll;
            }
        };
        // (the ring's first MIRROR_PX positions are kept twice: the lanes that hold them write the copy behind the ring's end, every other lane writes its own position again)
        auto seg_store = [&](int slot, const u32x4 (&h)[NQ], const u32x4 (&l)[NQ]) __attribute__((always_inline)) {
            unsigned char* p = ring + (slot * SEG_PX + lane) * PIXB;
            unsigned char* pm = p + (((int)(slot == 0) & (int)(lane < MIRROR_PX)) ? Rb : 0u);
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                *reinterpret_cast<u32x4*>(p + 16 * i) = h[i];
                *reinterpret_cast<u32x4*>(p + LO_OFF + 16 * i) = l[i];
                *reinterpret_cast<u32x4*>(pm + 16 * i) = h[i];
                *reinterpret_cast<u32x4*>(pm + LO_OFF + 16 * i) = l[i];
            }
        };
        __amdgpu_buffer_rsrc_t rs_out2 = rs_out;    // fused 1x1, channels-last: the image's (H W, 64) block
        if constexpr (FUSE == 2) rs_out2 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)b * 64 * HW), 0, (int)(64 * HW * sizeof(float)), 0x00020000);
        auto out2_voff = [&](int p) {              // fused 1x1: lane (position l & 15 of a 16-position half, couts 16 wave + 4 (l >> 4) + j)
            const int oy = row_of(p), ox = p - oy * P;
            const bool inside = (int)(oy < H) & (int)(ox < wv);
            const int off = FUSE == 2 ? ((oy * W + x0 + ox) * 64 + 16 * wave + 4 * (lane >> 4)) * 4 : ((4 * (lane >> 4)) * HW + oy * W + x0 + ox) * 4;
            return inside ? off : (int)0x80000000;
        };
        auto out_voff = [&](int p) {               // output position p of the padded raster -> this lane's store offset (its first cout), or "dropped"
            const int oy = row_of(p), ox = p - oy * P;
            const bool inside = (int)(oy < H) & (int)(ox < wv);
            return inside ? ((4 * kg) * HW + oy * W + x0 + ox) * 4 : (int)0x80000000;
        };
        // prologue: the window of the run's first unit (segments ua .. ua + nseg - 1: the whole ring) with every pipe idle; the segment the first unit will write travels
        float v[NV];
        u32x4 sh[NQ], sl[NQ];
        {
            constexpr int MAXN = C128 ? MAX_NSEG128 : FUSE ? 4 : MAX_NSEG;
            float vp[MAXN][NV];                     // every segment's loads travel together (one memory latency for the ring, not one per segment)
#pragma unroll
            for (int q = 0; q < MAXN; ++q) seg_load(ua + q, q < a.nseg, vp[q]);
#pragma unroll
            for (int q = 0; q < MAXN; ++q)
                if (q < a.nseg) {
                    seg_convert(vp[q], sh, sl);
                    seg_store(q, sh, sl);
                }
        }
        XFH_WAVE_SYNC();
        // the weights' home: 144 of the 216 registers in the accumulation half of the register file (next to the 64 accumulators).  Pinned HERE, behind the first run's ring
        // fill: their 54 loads were issued at the kernel's entry and travel together with the segments' (a pin right behind the loads makes the prologue two memory
        // latencies long instead of one; later runs find them pinned: an empty statement)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 2; ++q) XFH_AGPR(A[t][cb][q]);
        if constexpr (TRACE) { if (tr && n_units == 0) tr[2] = __builtin_amdgcn_s_memtime(); }
        if constexpr (!C128) seg_load(ua + a.nseg, ua + 1 < ub, v);
        int rslot = 0;                             // slot of segment u
        rowb = row_base(0u);
        ldx(rowb, std::integral_constant<int, 0>{}, x[0]);
        ldx(rowb, std::integral_constant<int, 1>{}, x[1], false);      // (tap 1's operands: pinned at the end of tap 0)
        // the segment's loads one by one (64 channels: two per slot of the second block, behind the conversion that empties their registers)
        int sv_off = (int)0x80000000;
        auto seg_voff = [&](int s, bool en) __attribute__((always_inline)) {
            const int i = 64 * s + lane, r = row_of(i), c = i - r * P;
            const int iy = r - 1, ix = x0 + c - 1;
            const bool inside = (int)en & (int)((unsigned)iy < (unsigned)H) & (int)((unsigned)ix < (unsigned)W);
            sv_off = inside ? (iy * W + ix) * 4 : (int)0x80000000;
        };
        auto seg_load_piece = [&](auto JC, float (&v)[NV]) __attribute__((always_inline)) {
            constexpr int j = decltype(JC)::value;
            v[j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_in, sv_off, j * HW * 4, 0));
        };
        auto seg_store_piece = [&](auto IC, int slot, const u32x4 (&h)[NQ], const u32x4 (&l)[NQ]) __attribute__((always_inline)) {
            constexpr int i = decltype(IC)::value;
            unsigned char* p = ring + (slot * SEG_PX + lane) * PIXB;
            unsigned char* pm = p + (((int)(slot == 0) & (int)(lane < MIRROR_PX)) ? Rb : 0u);
            *reinterpret_cast<u32x4*>(p + 16 * i) = h[i];
            *reinterpret_cast<u32x4*>(p + LO_OFF + 16 * i) = l[i];
            *reinterpret_cast<u32x4*>(pm + 16 * i) = h[i];
            *reinterpret_cast<u32x4*>(pm + LO_OFF + 16 * i) = l[i];
        };
        for (int u = ua; u < ub; ++u) {
            const unsigned t0b = (unsigned)(rslot * SEG_BYTES);
            // (128 channels: the next segment's 32 loads are issued here, not a unit earlier -- their registers would overlap the converted segment's for a whole unit,
            // and the maps of these layers sit in L2)
            RS_STAMP_U(3)
            if constexpr (C128) seg_load(u + a.nseg, u + 1 < ub, v);
            // ---- first block (accumulators a); in its slots: the reduction of the block before (accumulators b: the previous unit's, or run's, second block)
            if constexpr (FUSE == 0) { pend_a.rs = rs_out; pend_a.voff = out_voff(64 * u + n); }
            else { p2a_prev = p2a; p2a.rs = rs_out2; p2a.voff[0] = out2_voff(64 * u + (lane & 15)); p2a.voff[1] = out2_voff(64 * u + 16 + (lane & 15)); }
            {
                float4 part[3][2];
                float y[8];
                C1x1 w1;
                auto slot = [&](auto IDC) __attribute__((always_inline)) {
                    constexpr int id = decltype(IDC)::value;
                    RS_STAMP_U(32 + id)
                    if constexpr (id < NPIECE) red_write_piece(IDC, cb0, cb1);
                    if constexpr (id == 8) { RS_STAMP_U(4) }
                    if constexpr (id == 14) {
                        RS_STAMP_U(5)
                        XFH_LDS_BARRIER();
                        RS_STAMP_U(6)
                        red_read(part);
                        ++kb;
                    }
                    if constexpr (id == 15) {
                        if constexpr (FUSE == 0) finish_values(cb0, cb1, part, y);
                        else red_finish(cb0, cb1, part, pend_b, 1);
                    }
                    if constexpr (FUSE == 0 && id >= 17 && id < 17 + NSTORE) finish_store(std::integral_constant<int, id - 17>{}, y, pend_b);
                    if constexpr (FUSE != 0 && id >= 16 && id < 26) conv1x1_piece(IDC, w1, 0, p2a_prev);      // the first block of the unit before: its 3x3 outputs were published by this block's barrier
                };
                taps9(I0{}, t0b, ca0, ca1, x, slot);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) { cb0[i] = 0.f; cb1[i] = 0.f; }
            RS_STAMP_U(7)
            // ---- second block (accumulators b); in its slots: the first block's reduction, the conversion of the segment the NEXT unit needs (taps 0-2), the loads of the segment
            // after that (into the registers the conversion emptied), and behind the unit's last read of the ring the converted segment (over the segment this unit started with)
            if constexpr (FUSE == 0) { pend_b.rs = rs_out; pend_b.voff = out_voff(64 * u + 32 + n); }
            else { p2b_prev = p2b; p2b.rs = rs_out2; p2b.voff[0] = out2_voff(64 * u + 32 + (lane & 15)); p2b.voff[1] = out2_voff(64 * u + 48 + (lane & 15)); }
            {
                float4 part[3][2];
                float y[8];
                C1x1 w1;
                auto slot = [&](auto IDC) __attribute__((always_inline)) {
                    constexpr int id = decltype(IDC)::value;
                    RS_STAMP_U(32 + 27 + id)
                    if constexpr (id < NPIECE) red_write_piece(IDC, ca0, ca1);
                    if constexpr (id == 0) seg_convert(v, sh, sl);
                    if constexpr (id == 8) {
#pragma unroll
                        for (int i = 0; i < NQ; ++i) { XFH_VGPR(sh[i]); XFH_VGPR(sl[i]); }      // the conversion is complete here (left alone the compiler sinks it to the segment's store)
                        RS_STAMP_U(8)
                        if constexpr (!C128) seg_voff(u + 1 + a.nseg, u + 2 < ub);
                    }
                    if constexpr (!C128 && id >= 9 && id < 17) {
                        seg_load_piece(std::integral_constant<int, 2 * (id - 9)>{}, v);
                        seg_load_piece(std::integral_constant<int, 2 * (id - 9) + 1>{}, v);
                    }
                    if constexpr (id == 14) {
                        RS_STAMP_U(9)
                        XFH_LDS_BARRIER();
                        RS_STAMP_U(10)
                        red_read(part);
                        ++kb;
                    }
                    if constexpr (id == 15) {
                        if constexpr (FUSE == 0) finish_values(ca0, ca1, part, y);
                        else red_finish(ca0, ca1, part, pend_a, 0);
                    }
                    if constexpr (FUSE == 0 && id >= 17 && id < 17 + NSTORE) finish_store(std::integral_constant<int, id - 17>{}, y, pend_a);
                    if constexpr (FUSE != 0 && id >= 16 && id < 26) conv1x1_piece(IDC, w1, 1, p2b_prev);
                    // the ring's new segment: every operand read of this unit that touches the old one has been issued (tap 8 reads the next block's tap 0)
                    if constexpr (!C128 && (id == 24 || id == 25)) seg_store_piece(std::integral_constant<int, id - 24>{}, rslot, sh, sl);
                    if constexpr (C128 && id >= 23) seg_store_piece(std::integral_constant<int, C128 ? id - 23 : 0>{}, rslot, sh, sl);
                };
                taps9(I1{}, t0b + 32 * PIXB, cb0, cb1, x, slot);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) { ca0[i] = 0.f; ca1[i] = 0.f; }
            RS_STAMP_U(11)
            XFH_WAVE_SYNC();
            RS_STAMP_U(12)
            ++n_units;
            rslot = rslot + 1 == a.nseg ? 0 : rslot + 1;
        }
    }
    // ---- the last block of all
    {
        float4 part[3][2];
        red_write(cb0, cb1);
        XFH_LDS_BARRIER();
        red_read(part);
        red_finish(cb0, cb1, part, pend_b, 1);
        if constexpr (FUSE != 0) {
            XFH_LDS_BARRIER();
            conv1x1(0, p2a);
            conv1x1(1, p2b);
        }
    }
    fx_report_h(amax, a.status);
    fx_report(amaxf, a.status);
    RS_STAMP(13)
    if constexpr (TRACE) { if (tr) tr[14] = n_units; }
#undef RS_STAMP
#undef RS_STAMP_U
}

template <int FUSE, int CIN = 64, bool TRACE = false>      // FUSE 0: the 3x3 alone; 1: + trailing 1x1 (64 -> 64), NCHW output; 2: the same with channels-last output.  CIN 128: the 128 -> 128 layers (FUSE 0)
__device__ __forceinline__ void conv_rs64_body(const Rs64Args& a) {
    switch (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6)) {
        case 0: conv_rs64_wave<0, FUSE, CIN, TRACE>(a); break;
        case 1: conv_rs64_wave<1, FUSE, CIN, TRACE>(a); break;
        case 2: conv_rs64_wave<2, FUSE, CIN, TRACE>(a); break;
        default: conv_rs64_wave<3, FUSE, CIN, TRACE>(a); break;
    }
}

}  // namespace xfh
