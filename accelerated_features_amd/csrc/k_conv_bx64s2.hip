// 3x3 stride-2 convolution, 64 -> 64 or 64 -> 128 channels (block4.0 / block5.0; modules/model.py:68,75), on the bf16 matrix cores with
// three-way split operands (the arithmetic of k_conv_bx.hip; the weight stream and the input chunks of k_conv_bx64.hip).
//
// These two layers were the last direct convolutions on the f32 matrix cores (0.49 / 0.36 of THAT peak: 74 + 51 us per 64-frame step for
// 5.7 + 2.8 GFLOP); six bf16 MFMAs per K = 16 carry the same fp32 product sums at 2.7x the rate.  The maps are small (30x40 / 15x20
// outputs per VGA frame), so the kernel is shaped by balance, not by reuse:
//   * unit = (cout half, image, 16-column strip, 8-row tile): 8x16 output pixels x 64 couts.  VGA batch 64: 768 units for block4.0
//     (three per CU), 512 for block5.0 (two per CU: the second cout half of an image is another unit, not another accumulator);
//   * ONE workgroup of 8 waves per CU: wave (pb, cb) owns pixel block pb (2 output rows x 16 columns) and cout block cb: one 32x32
//     accumulator, per K step 3 + 3 ds_read_b128 for 6 MFMAs (half of the LDS read rate with two waves per SIMD);
//   * the 17x33 input halo of a tile goes through LDS in chunks of 16 channels with EVEN and ODD columns apart
//     ([17 rows, 3712 B apart][parity, 1904 B apart][17 / 16 pixels, 112 B apart][split h, m, l][16 channels] bf16): the 16 lanes of a
//     ds_read_b128 group step by two input pixels and would collide pairwise in one plane; 112 B keeps them on distinct banks;
//   * raw fp32 values are prefetched into registers one chunk ahead (also across units) as dwordx4 loads of pixel quads, split on
//     the way into LDS;
//   * the split weights (216 KiB per cout half) stream through a three-slot LDS ring by LDS-DMA, one slot = one tap row of one chunk
//     (3 K steps, 18 KiB); the DMA of row r + 2 is issued behind the barrier that opens row r.
#include "kernels.hpp"
#include "bx_split.hpp"

namespace xfh {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

struct Bx64S2Args {
    const float* in;
    const void* wq;            // [cout half][cin/16][3 dy][3 dx][2 cout blocks][3 splits][64 lanes][8 bf16]   (api.hip)
    const float* bias;
    float* out;
    int relu, H, W, Ho, Wo, B;
    int nrows, upi;            // 8-row tiles per strip, units per image and cout half
    long long* trace;          // debug: s_memtime stamps of the workgroup's second unit (NULL in production)
};

namespace bx64s2 {
constexpr int PIXB = 112, SPLB = 32, IH = 17, NEVEN = 17;
constexpr int PARB = NEVEN * PIXB;                      // odd columns of a row behind its even ones
constexpr int XROWB = 3712;                             // >= (17 + 16) * 112
constexpr int X_BYTES = IH * XROWB;                     // 63104
constexpr int STEP_BYTES = 2 * 3 * 1024, SLOT_BYTES = 3 * STEP_BYTES, NPIECE = SLOT_BYTES / 1024;
constexpr int NSLOT = 3;                                // ring depth: row r + 2 is in flight while row r is multiplied
constexpr int RING_OFF = X_BYTES, BIAS_OFF = RING_OFF + NSLOT * SLOT_BYTES, LDS_BYTES = BIAS_OFF + 128 * 4;
constexpr int NQ = 9;                                   // aligned 4-pixel quads [2 ox0 - 4, 2 ox0 + 32) per halo row
constexpr int NITEM = IH * NQ * 2;                      // (row, quad, 8-channel group)
static_assert(NITEM <= 512 && RING_OFF % 64 == 0, "one staging item per thread");
}

template <int NCO>          // cout halves: 1 (64 couts) or 2 (128)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_bx64s2_kernel(Bx64S2Args a) {
    using namespace bx64s2;
    constexpr int CIN = 64, NCH = CIN / 16, NROW = NCH * 3, COUT = 64 * NCO;
    static_assert(NROW % NSLOT == 0, "the ring slot of a row must not depend on the unit");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s2[];
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
    auto lds_addr = [](const unsigned char* p) { return (unsigned)(size_t)(lptr_t)p; };
    auto issue_row = [&](int r, int hf) __attribute__((always_inline)) {      // weights of row r (chunk r / 3, tap row r % 3) of cout half hf -> slot r % 3
        for (int j = wave; j < NPIECE; j += 8) {
            const unsigned m0v = lds_addr(smem_s2 + RING_OFF + (r % NSLOT) * SLOT_BYTES + j * 1024);
            const int soff = (hf * NROW + r) * SLOT_BYTES + j * 1024;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(dma_voff), "s"(rs_w), "s"(soff) : "memory");
        }
    };
    // Barrier that opens a row: everything but the DMA pieces of the row issued LAST has landed, for every wave.  vmcnt counts in issue
    // order, so "at most n outstanding" with n = this wave's pieces per row (3 for waves 0 and 1, 2 for the others) leaves only the newest
    // row in flight -- or less, if loads / stores were issued behind it (conservative).  (Measured: 72 -> 70 us for block4.0 against the
    // two-slot ring with a wait for everything.)
    auto dma_barrier = [&]() {
        if (wave < 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __syncthreads();
    };

    // ---- raw fp32 values of one 16-channel chunk of a tile: item of a thread = 4 consecutive pixels x 8 channels (eight dwordx4 loads,
    // one per channel plane).  Halo column c = 0 .. 32 is image column 2 ox0 - 1 + c; the quads start at 2 ox0 - 4.
    const bool has_item = tid < NITEM;
    const int it_g8 = tid / (IH * NQ), it_row = (tid - it_g8 * (IH * NQ)) / NQ, it_quad = tid % NQ;
    float v[8][4];
    int v_gx = 0;                             // first column of the quad in flight (W % 4 != 0: the tail of a quad that straddles the right border is
                                              // masked where it is consumed, k_conv_bx64.hip)
    auto issue_loads = [&](const Tile& t, int chunk) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in + (size_t)t.b * CIN * HW), 0, (int)(CIN * HW * sizeof(float)), 0x00020000);
        const int gy = 2 * t.oy0 - 1 + it_row, gx = 2 * t.ox0 - 4 + 4 * it_quad;
        const bool ok = has_item && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        v_gx = gx;
        const int voff = ok ? (int)((((size_t)it_g8 * 8) * HW + (size_t)gy * a.W + gx) * 4) : (int)0x80000000;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const auto q = __builtin_amdgcn_raw_buffer_load_b128(rs_in, voff, (int)((chunk * 16 + k) * HW * 4), 0);
            v[k][0] = __uint_as_float(q[0]); v[k][1] = __uint_as_float(q[1]); v[k][2] = __uint_as_float(q[2]); v[k][3] = __uint_as_float(q[3]);
        }
    };
    // split3 works on the two neighbouring PIXELS of a loaded quad; v_perm_b32 then gathers the channel pairs of each pixel (k_conv_bx64.hip)
    auto stage_write = [&]() __attribute__((always_inline)) {
        if (!has_item) return;
        unsigned char* row_base = smem_s2 + it_row * XROWB + it_g8 * 16 + 2 * it_quad * PIXB;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            unsigned H[8], M[8], L[8];                     // {pixel 2 pp, pixel 2 pp + 1} of channel k
            const bool z0 = (a.W & 3) && v_gx + 2 * pp >= a.W, z1 = (a.W & 3) && v_gx + 2 * pp + 1 >= a.W;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float x0 = v[k][2 * pp], x1 = v[k][2 * pp + 1];
                if (a.W & 3) { x0 = z0 ? 0.f : x0; x1 = z1 ? 0.f : x1; }
                split3(x0, x1, H[k], M[k], L[k]);
            }
#pragma unroll
            for (int e2 = 0; e2 < 2; ++e2) {
                const int e = 2 * pp + e2;                               // pixel of the quad: halo column c = 4 quad + e - 3
                if (it_quad == 0 && e < 3) continue;                     // (c < 0: left of the halo)
                const int par = (e + 1) & 1;                             // c & 1
                const int idx = e == 0 ? -2 : e == 3 ? 0 : -1;           // (c >> 1) - 2 quad
                const unsigned sel = e2 ? 0x07060302u : 0x05040100u;
                uint4 h, m, l;
                h.x = __builtin_amdgcn_perm(H[1], H[0], sel); h.y = __builtin_amdgcn_perm(H[3], H[2], sel);
                h.z = __builtin_amdgcn_perm(H[5], H[4], sel); h.w = __builtin_amdgcn_perm(H[7], H[6], sel);
                m.x = __builtin_amdgcn_perm(M[1], M[0], sel); m.y = __builtin_amdgcn_perm(M[3], M[2], sel);
                m.z = __builtin_amdgcn_perm(M[5], M[4], sel); m.w = __builtin_amdgcn_perm(M[7], M[6], sel);
                l.x = __builtin_amdgcn_perm(L[1], L[0], sel); l.y = __builtin_amdgcn_perm(L[3], L[2], sel);
                l.z = __builtin_amdgcn_perm(L[5], L[4], sel); l.w = __builtin_amdgcn_perm(L[7], L[6], sel);
                unsigned char* p = row_base + par * PARB + idx * PIXB;
                *reinterpret_cast<uint4*>(p) = h;
                *reinterpret_cast<uint4*>(p + SPLB) = m;
                *reinterpret_cast<uint4*>(p + 2 * SPLB) = l;
            }
        }
    };

    long long* tr = a.trace && tid == 0 ? a.trace + (size_t)blockIdx.x * 64 : nullptr;
    int tix = 0;
#define S2_STAMP(k) { if (tr && tix == 1) tr[k] = __builtin_amdgcn_s_memtime(); }      /* [0] unit start; row r: [1+4r] start, [2+4r] barrier passed, [3+4r] MFMAs issued; [50] stores issued, [51] end barrier */
    struct Frag { bf16x8 x[3]; bf16x8 w[3]; };
    // lane (pixel l31 of block pb): output row 2 pb + (l31 >> 4), column l31 & 15 -> input row 2 * that (+ dy), even column index = column (+ dx >> 1)
    const int lane_px = 2 * (2 * pb + (l31 >> 4)) * XROWB + (l31 & 15) * PIXB + half * 16;

    auto do_tile = [&](const Tile& cur, const Tile& nxt, bool has_next) __attribute__((always_inline)) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int c = 0; c < NCH; ++c) {
            if (c > 0) dma_barrier();          // every wave has finished the previous chunk's last tap row (the unit loop ends on a barrier)
            stage_write();
            for (int dy = 0; dy < 3; ++dy) {
                const int r = c * 3 + dy;
                S2_STAMP(1 + 4 * r)
                dma_barrier();                 // row r's weights landed; (dy = 0) the chunk is staged; (dy > 0) row r - 1 is finished
                S2_STAMP(2 + 4 * r)
                const unsigned char* wslot = smem_s2 + RING_OFF + (r % NSLOT) * SLOT_BYTES + cb * 3 * 1024 + lane * 16;
                const unsigned char* xrow = smem_s2 + lane_px + dy * XROWB;
                Frag f[2];
                auto load = [&](int s, Frag& o) {          // tap column s: parity s & 1, pixel index + (s >> 1)
#pragma unroll
                    for (int q = 0; q < 3; ++q) o.x[q] = *reinterpret_cast<const bf16x8*>(xrow + (s & 1) * PARB + (s >> 1) * PIXB + q * SPLB);
#pragma unroll
                    for (int q = 0; q < 3; ++q) o.w[q] = *reinterpret_cast<const bf16x8*>(wslot + s * STEP_BYTES + q * 1024);
                };
                load(0, f[0]);
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const Frag& cf = f[s & 1];
                    if (s + 1 < 3) load(s + 1, f[(s + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    // products (weight split, input split), small terms first: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cf.w[2], cf.x[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cf.w[0], cf.x[2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cf.w[1], cf.x[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cf.w[1], cf.x[0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cf.w[0], cf.x[1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cf.w[0], cf.x[0], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // memory instructions BETWEEN the MFMA groups, behind idle slots (VALU address arithmetic right behind an MFMA may land in
                    // operand lanes it has not read yet)
                    if (s < 2) { asm volatile("s_nop 7\n\ts_nop 7"); __builtin_amdgcn_sched_barrier(0); }
                    if (s == 0) issue_row(r + 2 < NROW ? r + 2 : r + 2 - NROW, r + 2 < NROW ? cur.hf : nxt.hf);      // (the stream is cyclic over the units)
                    if (s == 1 && dy == 0) {   // raw values of the next chunk (or the next unit's first) fly under this chunk's MFMAs (ONE load site)
                        const bool same = c + 1 < NCH;
                        Tile lt;
                        lt.b = same ? cur.b : nxt.b; lt.oy0 = same ? cur.oy0 : nxt.oy0; lt.ox0 = same ? cur.ox0 : nxt.ox0; lt.hf = 0;
                        if (same || has_next) issue_loads(lt, same ? c + 1 : 0);
                    }
                }
                asm volatile("s_nop 7\n\ts_nop 7");
                __builtin_amdgcn_sched_barrier(0);
                S2_STAMP(3 + 4 * r)
            }
        }
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
            float y = acc[r] + bs[r];
            if (a.relu) y = fmaxf(y, 0.f);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y), rs_out, voff, (int)(((r & 3) + 8 * (r >> 2)) * HWo * 4), 0);
        }
    };

    Tile cur, nxt;
    int u = u0;
    tile_at(u++, cur);
    nxt = cur;
    issue_row(0, cur.hf);
    issue_row(1, cur.hf);
    issue_loads(cur, 0);
    for (;;) {
        const bool has_next = u < u1;
        if (has_next) tile_at(u++, nxt);
        S2_STAMP(0)
        do_tile(cur, nxt, has_next);
        S2_STAMP(50)
        if (!has_next) break;
        dma_barrier();                         // every wave is done with the unit's last tap row before the next chunk is staged
        S2_STAMP(51)
        ++tix;
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the cyclic stream's last DMA must not outlive the workgroup's LDS
#undef S2_STAMP
}

template <int NCO>
static int run_bx64s2(const ConvW& c, const float* in, int B, int H, int W, float* out, hipStream_t st, long long* trace) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if ((size_t)64 * H * W * sizeof(float) >= 0x7fffffffu) return -1;      // buffer-resource range
    Bx64S2Args a;
    a.in = in; a.wq = c.w_bx; a.bias = c.bias; a.out = out; a.relu = c.relu; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.B = B; a.trace = trace;
    a.nrows = ceil_div(Ho, 8); a.upi = ceil_div(Wo, 16) * a.nrows;
    static unsigned attr_done = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(conv_bx64s2_kernel<NCO>), bx64s2::LDS_BYTES, attr_done);
    const long long units = (long long)NCO * B * a.upi;
    int grid = num_cus();                      // one 8-wave workgroup per CU (100 KiB of LDS); a multiple of 8 keeps a workgroup on its XCD
    if (units < grid) grid = (int)units;       // (small inputs: one unit per workgroup; the XCD mapping then needs grid % 8 == 0 or is skipped)
    conv_bx64s2_kernel<NCO><<<grid, 512, bx64s2::LDS_BYTES, st>>>(a);
    return 0;
}

int launch_conv_bx64s2(const ConvW& c, const float* in, int B, int H, int W, float* out, hipStream_t st, long long* trace) {
    if (c.ks != 3 || c.stride != 2 || !c.w_bx || c.cin != 64) return -1;
    if (c.cout == 64) return run_bx64s2<1>(c, in, B, H, W, out, st, trace);
    if (c.cout == 128) return run_bx64s2<2>(c, in, B, H, W, out, st, trace);
    return -1;
}

}  // namespace xfh
