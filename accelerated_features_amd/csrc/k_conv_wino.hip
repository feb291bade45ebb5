// 3x3 stride-1 convolution as Winograd F(2x2,3x3) on the f32 matrix cores.
//
// Why: the 3x3/s1 layers with >= 24 channels (block2.x, block3.1, block4.1/.2, block5.1/.2,
// block_fusion.0/.1; modules/model.py:50-77) are f32-MFMA bound in the direct implicit GEMM
// (k_conv_mfma.hip, which already issues one MFMA per ~77 cycles of a 64-cycle pipe) and carry
// ~85 % of the backbone's matrix work.  F(2x2,3x3) computes a 2x2 output tile from a 4x4 input
// patch with 16 multiplies per (cin,cout) instead of 36: 2.25x less matrix work, still plain
// fp32 arithmetic (the transforms are additions; the weight transform G g G^T is done once on
// the host in fp64 and rounded to fp32).
//
//   V[xi][nu][c][tile]  = (B^T d B)[xi][nu]          input transform, per channel and tile
//   M[xi][nu][co][tile] = sum_c U[xi][nu][co][c] * V[xi][nu][c][tile]     16 independent GEMMs
//   Y[2x2][co][tile]    = A^T M A                    output transform
//
// What bounds this kernel is the LDS, not the matrix pipe (a first version that staged V in LDS
// spent 1700-2000 LDS cycles per 2048 MFMA cycles and ran at 3300 cycles per 4-channel chunk).
// Hence the decomposition:
//   * wave (g, nu) owns ONE column nu of the 4x4 transform domain (4 positions xi = 0..3) for two
//     32x32 blocks (2 cout blocks x 1 tile block, or 1 x 2): 8 accumulators = 128 registers.
//   * the B operand (transformed input) never touches LDS: MFMA lane (tile j, k-parity) computes
//     v[xi][nu] for ITS tile and ITS channel straight from the raw tile -- 8 LDS dwords and
//     8 VALU ops per (tile, channel): column combine d[.][ca] +- d[.][cb], then the row transform.
//   * the raw tile is stored de-interleaved by column parity with a padded half-width, so those
//     dword reads are bank-conflict free; the A operand is read as float2 (both k-pairs of a chunk).
//   * raw tiles and transformed weights arrive by buffer_load ... lds (DMA, out-of-image lanes carry
//     an out-of-range offset: the hardware writes the zero padding), issued from inline asm: hipcc
//     makes every LDS read it can see wait for ALL LDS-DMA it knows of (vmcnt(0) in the middle of
//     the loop); asm DMA plus explicit waits before the barriers keeps the copy of chunk i+2 in
//     flight under the MFMAs of chunk i.
//   * per 4-channel chunk: one barrier; 16 MFMAs per wave; the operands of chunk i+1 are built
//     BETWEEN this wave's own MFMAs (sched_barrier pins); the two waves of a SIMD run complementary
//     schedules (one has its side work in the first half, the other in the second half).
//   * output transform: A^T M over xi is lane-local; the sum over nu crosses the 4 waves of a
//     group through LDS once; wave nu finishes output row i = nu&1 of block nu>>1 and stores float2.
#include "kernels.hpp"
#include <cstdlib>
#include <type_traits>

namespace xfh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

struct WinoArgs {
    const float* in;
    const float* wu;          // [CIN/4][16 pos][2 half][COUT_PAD][2 p]   channel = 4*chunk + 2*p + half
    const float* bias;        // [COUT_PAD]
    float* out;
    const float* wk2;         // fused 1x1: [COUT][COUT2_PAD]
    const float* bias2;       // [COUT2_PAD]
    int relu, relu2;
    int H, W, B;              // input == output size (stride 1, pad 1)
    int tiles_x, tiles;       // workgroups per image
    long long* trace;         // debug: per-workgroup s_memtime stamps (NULL in production)
    int cold;
};

// smallest padded half-width >= ttw+1 for which the 32 lanes of a half-wave (tiles t = 0..31, tile
// (ty,tx) at dword (4*hwp)*ty + tx of a channel plane) hit 32 different banks
constexpr bool wino_bank_ok(int hwp, int ttw) {
    unsigned seen = 0;
    for (int t = 0; t < 32; ++t) {
        const int bank = ((t / ttw) * 4 * hwp + t % ttw) & 31;
        if (seen & (1u << bank)) return false;
        seen |= 1u << bank;
    }
    return true;
}
constexpr int wino_pick_hwp(int ttw) {
    for (int h = ttw + 1; h < ttw + 33; ++h)
        if (wino_bank_ok(h, ttw)) return h;
    return ttw + 1;
}

template <int CB, int TBG, int NCBW, int NTBW, int TTH_, int TTW_, int COUT2 = 0>
struct WinoCfg {
    static constexpr int CK = 4, TTH = TTH_, TTW = TTW_;
    static constexpr int NG = (CB / NCBW) * (TBG / NTBW), NW = 4 * NG, NTHR = 64 * NW;
    static constexpr int COUT_PAD = 32 * CB, NT = 32 * TBG;
    static constexpr int HWP = wino_pick_hwp(TTW), ROWS = 2 * TTH + 2, PLANE = ROWS * 2 * HWP;
    static constexpr int PS = (PLANE + 64 * NW - 1) / (64 * NW) * (64 * NW), NSEG = PS / (64 * NW);
    static constexpr int UCH = 16 * CK * COUT_PAD, UPW = UCH / 256 / NW;
    static constexpr int RING = 2 * UCH + 2 * CK * PS;           // floats: two slots each
    static constexpr int XCH = NW * 4 * 16 * 64;                 // floats: output-transform exchange
    static constexpr int COUT2_PAD = (COUT2 + 31) / 32 * 32;
    static constexpr int W2_OFF = RING > XCH ? RING : XCH;      // fused 1x1 weights live behind the ring / exchange area
    static constexpr int LDS_FLOATS = W2_OFF + 32 * CB * COUT2_PAD;
    static_assert(NCBW * NTBW == 2 && CB % NCBW == 0 && TBG % NTBW == 0, "two 32x32 blocks per wave");
    static_assert(TTH * TTW <= NT && TTH * TTW > NT - 32, "region must fill the tile blocks");
    static_assert((UCH / 256) % NW == 0, "every wave issues the same number of weight DMAs");
};

template <int CIN, int COUT, int CB, int TBG, int NCBW, int NTBW, int TTH, int TTW, int COUT2, bool NHWC>
__global__ __launch_bounds__(256 * (CB / NCBW) * (TBG / NTBW)) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_wino_kernel(WinoArgs a) {
    kernel_entry_hooks(a.cold);      // debug: code-position shift / cold instruction cache (common.hpp)
    using Cfg = WinoCfg<CB, TBG, NCBW, NTBW, TTH, TTW, COUT2>;
    constexpr int COUT2_PAD = Cfg::COUT2_PAD, MB2 = COUT2_PAD / 32;
    static_assert(COUT2 == 0 || (COUT == 32 * CB && NCBW == 2 && MB2 == 2), "fused 1x1: full cout blocks, two per wave, 64 outputs");
    static_assert(COUT2 > 0 || !NHWC, "channels-last output only with the fused 1x1");
    constexpr int CK = Cfg::CK, NW = Cfg::NW, COUT_PAD = Cfg::COUT_PAD, HWP = Cfg::HWP, ROWS = Cfg::ROWS;
    constexpr int PS = Cfg::PS, NSEG = Cfg::NSEG, UCH = Cfg::UCH, UPW = Cfg::UPW, NCH = CIN / CK;
    static_assert(CIN % (2 * CK) == 0, "the main loop is unrolled by two chunks");
    static_assert(COUT <= COUT_PAD && COUT_PAD - COUT < 32, "CB must match COUT");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ul = smem;                 // [2][UCH]
    float* Rl = smem + 2 * UCH;       // [2][CK][PS]
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nu = wave & 3, g = wave >> 2;          // waves w and w+4 share a SIMD: same nu, different group
    const int cb0 = (g % (CB / NCBW)) * NCBW, tb0 = (g / (CB / NCBW)) * NTBW;
    // per-tile state
    int b = 0, oy0 = 0, ox0 = 0;
    const size_t HW = (size_t)a.H * a.W;
    const int HWb = (int)(HW * sizeof(float));
    long long* tr = a.trace ? a.trace + (size_t)blockIdx.x * 24 : nullptr;
    if (tr && tid == 0) { tr[0] = __builtin_amdgcn_s_memtime(); tr[19] = __builtin_amdgcn_s_memrealtime(); }

    // ---- DMA (inline asm: see header) ----------------------------------------------------------
    auto make_rsrc = [](const void* p, unsigned bytes) {
        const unsigned long long ba = (unsigned long long)p;
        i32x4 r;
        // readfirstlane: the "s" asm constraint needs values the compiler KNOWS to be wave-uniform
        r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ba);
        r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(ba >> 32) & 0xffffu));
        r.z = __builtin_amdgcn_readfirstlane((int)bytes);
        r.w = 0x00020000;
        return r;
    };
    i32x4 rs_in;
    const i32x4 rs_u = make_rsrc(a.wu, (unsigned)((size_t)CIN * 16 * COUT_PAD * sizeof(float)));
    // raw plane element e of a channel = (row r, column parity q, half column h): source pixel (r, 2h+q)
    int xvoff[NSEG];
    auto set_tile = [&](int vid) {       // false for the padding ids of the XCD-swizzled grid
        int tile;
        if (!xcd_group_map(vid, a.tiles, a.B, b, tile)) return false;
        const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
        oy0 = 2 * tyi * TTH; ox0 = 2 * txi * TTW;
        rs_in = make_rsrc(a.in + (size_t)b * CIN * HW, (unsigned)(CIN * HW * sizeof(float)));
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
            const int e = (wave + NW * s) * 64 + lane;
            const int r = e / (2 * HWP), rem = e - r * (2 * HWP), q = rem / HWP, h = rem - q * HWP;
            const int gy = oy0 - 1 + r, gx = ox0 - 1 + 2 * h + q;
            const bool ok = r < ROWS && h <= TTW && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
            xvoff[s] = ok ? (gy * a.W + gx) * 4 : (int)0x80000000;
        }
        return true;
    };
    if (!set_tile(blockIdx.x)) return;
    const int uvoff = lane * 16;
    auto lds_addr = [](const float* p) { return (unsigned)(size_t)(lptr_t)p; };
    auto issue = [&](int ch, int slot) {
#pragma unroll
        for (int jj = 0; jj < UPW; ++jj) {
            const int j = wave + NW * jj;
            const unsigned m0v = lds_addr(Ul + slot * UCH + j * 256);
            const int soff = (ch * UCH + j * 256) * 4;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(uvoff), "s"(rs_u), "s"(soff) : "memory");
        }
        i32x4 rin;                     // per-tile resource: re-assert uniformity where the "s" operand is formed
        rin.x = __builtin_amdgcn_readfirstlane(rs_in.x); rin.y = __builtin_amdgcn_readfirstlane(rs_in.y);
        rin.z = __builtin_amdgcn_readfirstlane(rs_in.z); rin.w = 0x00020000;
#pragma unroll
        for (int s = 0; s < NSEG; ++s)
#pragma unroll
            for (int c = 0; c < CK; ++c) {
                const unsigned m0v = lds_addr(Rl + (slot * CK + c) * PS + (wave + NW * s) * 64);
                const int soff = (ch * CK + c) * HWb;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(m0v), "v"(xvoff[s]), "s"(rin), "s"(soff) : "memory");
            }
    };
    if constexpr (COUT2 > 0) {       // fused 1x1 weights: one DMA, covered by the first barrier
        const i32x4 rs_w2 = make_rsrc(a.wk2, (unsigned)(COUT * COUT2_PAD * sizeof(float)));
        for (int j = wave; j < COUT * COUT2_PAD / 256; j += NW) {
            const unsigned m0v = lds_addr(smem + Cfg::W2_OFF + j * 256);
            const int soff = j * 1024;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(uvoff), "s"(rs_w2), "s"(soff) : "memory");
        }
    }
    auto dma_barrier = [&]() {       // everything this workgroup has in flight has landed, for every wave
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

    // ---- operand addressing -----------------------------------------------------------------------
    // B: this lane's tile (per tile block) and channel parity; column pair / sign of transform column nu
    const int ca = nu == 0 ? 0 : (nu == 2 ? 2 : 1), cb2 = nu == 2 ? 1 : (nu == 3 ? 3 : 2);
    const float csgn = nu == 1 ? 1.f : -1.f;
    int baseA[NTBW], baseB[NTBW];
#pragma unroll
    for (int j = 0; j < NTBW; ++j) {
        int t = (tb0 + j) * 32 + l31;
        if (t >= TTH * TTW) t = 0;
        const int ty = t / TTW, tx = t - ty * TTW;
        const int toff = half * PS + ty * 4 * HWP + tx;
        baseA[j] = toff + (ca & 1) * HWP + (ca >> 1);
        baseB[j] = toff + (cb2 & 1) * HWP + (cb2 >> 1);
    }
    // A: float2 {p=0, p=1} at [pos = 4*xi + nu][half][cout][p]
    const int abase = ((nu * 2 + half) * COUT_PAD + cb0 * 32 + l31) * 2;

    struct Ops {
        float2 a[NCBW][4];        // [cout block][xi] -> {k-pair 0, k-pair 1}
        float b[NTBW][2][4];      // [tile block][k-pair][xi]
    };
    auto load_a = [&](int slot, Ops& o) {
        const float* U = Ul + slot * UCH + abase;
#pragma unroll
        for (int c = 0; c < NCBW; ++c)
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) o.a[c][xi] = *reinterpret_cast<const float2*>(U + (xi * 8 * COUT_PAD + c * 32) * 2);
    };
    auto read_b = [&](int slot, int p, float (&d)[NTBW][2][4]) {
        const float* R = Rl + slot * CK * PS + 2 * p * PS;
#pragma unroll
        for (int j = 0; j < NTBW; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d[j][0][r] = R[baseA[j] + r * 2 * HWP];
                d[j][1][r] = R[baseB[j] + r * 2 * HWP];
            }
    };
    auto make_b = [&](int p, const float (&d)[NTBW][2][4], Ops& o) {
#pragma unroll
        for (int j = 0; j < NTBW; ++j) {
            float u[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) u[r] = __fmaf_rn(csgn, d[j][1][r], d[j][0][r]);
            o.b[j][p][0] = u[0] - u[2];
            o.b[j][p][1] = u[1] + u[2];
            o.b[j][p][2] = u[2] - u[1];
            o.b[j][p][3] = u[1] - u[3];
        }
    };

    auto body = [&](auto SCHED) {
        constexpr int SV = decltype(SCHED)::value;
        // Wave nu finishes output row oi = nu&1 of block blk = nu>>1 (both columns j: float2).
        const int oi = nu & 1, blk = nu >> 1;
        const int cbk = NCBW == 2 ? cb0 + blk : cb0, tbk = NTBW == 2 ? tb0 + blk : tb0;
        issue(0, 0);
        issue(1, 1);
        {
        f32x16 acc[4][2];          // [xi][block], block = cout-block-major
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[xi][k][r] = 0.f;
        dma_barrier();
        if (tr && tid == 0) tr[1] = __builtin_amdgcn_s_memtime();
        Ops ops[2];
        {
            float d[NTBW][2][4];
            load_a(0, ops[0]);
            read_b(0, 0, d); make_b(0, d, ops[0]);
            read_b(0, 1, d); make_b(1, d, ops[0]);
        }

#define XFH_PIN __builtin_amdgcn_sched_barrier(0)
        // one chunk: MFMAs on ops[CUR] (chunk i), operands of chunk i+1 -> ops[CUR^1], DMA of chunk i+2
        auto chunk = [&](int i, auto CURC) {
            constexpr int CUR = decltype(CURC)::value, NXT = CUR ^ 1;
            dma_barrier();           // chunk i+1 landed in slot NXT; every wave is done with slot CUR
            if (tr && tid == 0 && i < 8) tr[2 + i] = __builtin_amdgcn_s_memtime();
            const int c2 = i + 2 < NCH ? i + 2 : NCH - 1;      // past the end: re-fetch into a dead slot
            float d[NTBW][2][4];
#define M(m) { XFH_PIN; { constexpr int p_ = (m) >> 3, xi_ = ((m) >> 1) & 3, k_ = (m) & 1; constexpr int c_ = NCBW == 2 ? k_ : 0, t_ = NTBW == 2 ? k_ : 0; \
                 acc[xi_][k_] = __builtin_amdgcn_mfma_f32_32x32x2f32(p_ ? ops[CUR].a[c_][xi_].y : ops[CUR].a[c_][xi_].x, ops[CUR].b[t_][p_][xi_], acc[xi_][k_], 0, 0, 0); } XFH_PIN; }
#define P_DMA { issue(c2, CUR); }
#define P_A   { load_a(NXT, ops[NXT]); }
#define P_BR(p) { read_b(NXT, p, d); }
#define P_BV(p) { make_b(p, d, ops[NXT]); }
            if (SV == 0) {
                M(0) P_DMA M(1) P_A P_BR(0) M(2) M(3) M(4) P_BV(0) M(5) P_BR(1) M(6) M(7) M(8) P_BV(1)
                M(9) M(10) M(11) M(12) M(13) M(14) M(15)
            } else {
                M(0) M(1) M(2) M(3) M(4) M(5) M(6) P_DMA M(7) P_A P_BR(0) M(8) M(9) M(10) P_BV(0) M(11) P_BR(1)
                M(12) M(13) M(14) P_BV(1) M(15)
            }
#undef M
#undef P_DMA
#undef P_A
#undef P_BR
#undef P_BV
        };
        for (int i = 0; i < NCH; i += 2) {
            chunk(i, std::integral_constant<int, 0>{});
            chunk(i + 1, std::integral_constant<int, 1>{});
        }
#undef XFH_PIN
        if (tr && tid == 0) tr[20] = __builtin_amdgcn_s_memtime();

        // ---- output transform --------------------------------------------------------------------
        // T[i][blk] = sum_xi At[i][xi] M[xi]   (lane-local);  Y[i][j] = sum_nu T_nu[i] A[nu][j].
        float bs[16];                // bias: loaded here, in flight under the exchange
#pragma unroll
        for (int r = 0; r < 16; ++r) bs[r] = a.bias[cbk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
        {
            dma_barrier();               // all waves done with the rings (and the dummy tail DMA has landed)
            if (tr && tid == 0) tr[10] = __builtin_amdgcn_s_memtime();
            float* X = smem;             // [wave][i*2+k][16 r][64 lanes]
            f32x16 own;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const f32x16 t0 = acc[0][k] + acc[1][k] + acc[2][k];
                const f32x16 t1 = acc[1][k] - acc[2][k] - acc[3][k];
                if (k == blk) own = oi ? t1 : t0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (!(k == blk && oi == 0)) X[((wave * 4 + 0 + k) * 16 + r) * 64 + lane] = t0[r];
                    if (!(k == blk && oi == 1)) X[((wave * 4 + 2 + k) * 16 + r) * 64 + lane] = t1[r];
                }
            }
            if (tr && tid == 0) tr[11] = __builtin_amdgcn_s_memtime();
            dma_barrier();
            if (tr && tid == 0) tr[12] = __builtin_amdgcn_s_memtime();
            const float* Xg = X + ((g * 4) * 4 + oi * 2 + blk) * 16 * 64 + lane;      // + nu' * 4*16*64
            const int t = tbk * 32 + l31;
            const int ty = t / TTW, tx = t - ty * TTW;
            const int oy = oy0 + 2 * ty + oi, ox = ox0 + 2 * tx;
            const bool ok = t < TTH * TTW && oy < a.H && ox < a.W;
            const bool pair = ox + 1 < a.W;
            float* op = a.out + ((size_t)b * COUT * a.H + oy) * a.W + ox;
            const bool al8 = ((a.W & 1) == 0);          // ox is even: rows are 8-byte aligned iff W is even
            // coefficients of T_nu in (Y[.][0], Y[.][1]): nu 0: (1,0)  1: (1,1)  2: (1,-1)  3: (0,-1)
            float y0[16], y1[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                y0[r] = nu == 3 ? 0.f : own[r];
                y1[r] = nu == 0 ? 0.f : (nu == 1 ? own[r] : -own[r]);
            }
#pragma unroll
            for (int n2 = 0; n2 < 4; ++n2) {
                if (n2 == nu) continue;          // wave-uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = Xg[(n2 * 64 + r) * 64];
                    if (n2 != 3) y0[r] += v;
                    if (n2 == 1) y1[r] += v;
                    if (n2 >= 2) y1[r] -= v;
                }
            }
            if (tr && tid == 0) tr[13] = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                y0[r] += bs[r]; y1[r] += bs[r];
                if (a.relu) { y0[r] = fmaxf(y0[r], 0.f); y1[r] = fmaxf(y1[r], 0.f); }
            }
            if constexpr (COUT2 == 0) {
                if (ok) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = cbk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (co < COUT) {
                            float* o = op + (size_t)co * HW;
                            if (pair && al8) *reinterpret_cast<float2*>(o) = make_float2(y0[r], y1[r]);
                            else { o[0] = y0[r]; if (pair) o[1] = y1[r]; }
                        }
                    }
                }
            } else {
                // ---- fused trailing 1x1 (block3.1+3.2, block5.2+5.3, block_fusion.1+.2) --------------------
                // Register r of y holds, for this lane's tile, channel cbk*32 + (r&3)+8(r>>2) + 4*half: pairing
                // channels (c, c+4) makes y[r] THE B operand (A for channels-last) of the 1x1 GEMM -- no LDS
                // round trip for the activations.  This wave covers the K slice of its cout block cbk; the
                // KW = CB waves sharing (output row, tile block) add their partial sums through LDS.
                const float* W2l = smem + Cfg::W2_OFF;
                f32x16 acc2[2][2];         // [m2][pixel column j]
#pragma unroll
                for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc2[m2][j][r] = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kc = cbk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float w0 = W2l[kc * COUT2_PAD + l31], w1 = W2l[kc * COUT2_PAD + 32 + l31];
                    if (NHWC) {
                        acc2[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(y0[r], w0, acc2[0][0], 0, 0, 0);
                        acc2[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(y1[r], w0, acc2[0][1], 0, 0, 0);
                        acc2[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(y0[r], w1, acc2[1][0], 0, 0, 0);
                        acc2[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(y1[r], w1, acc2[1][1], 0, 0, 0);
                    } else {
                        acc2[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, y0[r], acc2[0][0], 0, 0, 0);
                        acc2[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w0, y1[r], acc2[0][1], 0, 0, 0);
                        acc2[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, y0[r], acc2[1][0], 0, 0, 0);
                        acc2[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1, y1[r], acc2[1][1], 0, 0, 0);
                    }
                }
                // K-split reduction: KW waves (kw = index of this wave's cout block) hold partial sums of the same
                // outputs.  Vector v = m2*2 + j is finished by wave kw = v*KW/4 (KW = 2: one m2, both columns).
                constexpr int KW = CB, VPW = 4 / KW;
                static_assert(KW == 2 || KW == 4, "K split over 2 or 4 waves");
                const int kw = cbk;                         // 0 .. KW-1
                dma_barrier();                              // every wave has finished reading the first exchange
                float* X2 = smem;                           // [wave][v][16 r][64 lanes]
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    if (v / VPW == kw) continue;            // wave-uniform: own vectors stay in registers
#pragma unroll
                    for (int r = 0; r < 16; ++r) X2[((wave * 4 + v) * 16 + r) * 64 + lane] = acc2[v >> 1][v & 1][r];
                }
                dma_barrier();
                float bs2[16];
                const int m2o = (kw * VPW) >> 1;            // the m2 block this wave finishes
#pragma unroll
                for (int r = 0; r < 16; ++r) bs2[r] = NHWC ? a.bias2[m2o * 32 + l31] : a.bias2[m2o * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
#pragma unroll
                for (int vv = 0; vv < VPW; ++vv) {
                    const int v = kw * VPW + vv;            // wave-uniform
                    f32x16 sum;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum[r] = 0.f;
#pragma unroll
                    for (int k2 = 0; k2 < KW; ++k2) {        // peers: same output row oi and tile block, cout block k2
                        const int pw = NCBW == 2 && CB == 2 ? (g * 4 + oi + 2 * k2) : ((tb0 / NTBW * (CB / NCBW) + (k2 >> 1)) * 4 + oi + 2 * (k2 & 1));
                        if (k2 == kw) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) sum[r] += (v == 0 ? acc2[0][0][r] : v == 1 ? acc2[0][1][r] : v == 2 ? acc2[1][0][r] : acc2[1][1][r]);
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; ++r) sum[r] += X2[((pw * 4 + v) * 16 + r) * 64 + lane];
                        }
                    }
                    const int jcol = v & 1;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float o2 = sum[r] + bs2[r];
                        if (a.relu2) o2 = fmaxf(o2, 0.f);
                        if (!NHWC) {
                            const int c2 = m2o * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            if (ok && c2 < COUT2 && ox + jcol < a.W) a.out[(((size_t)b * COUT2 + c2) * a.H + oy) * a.W + ox + jcol] = o2;
                        } else {
                            const int t2 = tbk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;      // D rows are tiles
                            const int ty2 = t2 / TTW, tx2 = t2 - ty2 * TTW;
                            const int oy2 = oy0 + 2 * ty2 + oi, ox2 = ox0 + 2 * tx2 + jcol;
                            const int c2 = m2o * 32 + l31;
                            if (t2 < TTH * TTW && oy2 < a.H && ox2 < a.W && c2 < COUT2) a.out[(((size_t)b * a.H + oy2) * a.W + ox2) * COUT2 + c2] = o2;
                        }
                    }
                }
            }
            if (tr && tid == 0) { tr[21] = __builtin_amdgcn_s_memtime(); tr[23] = __builtin_amdgcn_s_memrealtime(); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); tr[22] = ((long long)xcc << 32) | hwid; }
        }
        }
    };
    // the two waves of a SIMD (g even / odd) run complementary schedules
    if ((g & 1) == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
template <int CIN, int COUT, int CB, int TBG, int NCBW, int NTBW, int TTH, int TTW, int COUT2 = 0, bool NHWC = false>
static int run_wino(const ConvW& c, const float* in, int B, int H, int W, float* out, hipStream_t st, long long* trace, const ConvW* c2 = nullptr) {
    using Cfg = WinoCfg<CB, TBG, NCBW, NTBW, TTH, TTW, COUT2>;
    if ((size_t)c.cin * H * W * sizeof(float) >= 0x7fffffffu) return -1;     // buffer-resource range
    WinoArgs a;
    a.cold = g_debug_cold;
    a.in = in; a.wu = c.w_wino; a.bias = c.bias; a.out = out; a.relu = c.relu; a.H = H; a.W = W; a.B = B; a.trace = trace;
    a.wk2 = c2 ? c2->w_kcp : nullptr; a.bias2 = c2 ? c2->bias : nullptr; a.relu2 = c2 ? c2->relu : 0;
    if ((COUT2 > 0) != (c2 != nullptr)) return -1;
    a.tiles_x = ceil_div(ceil_div(W, 2), TTW);
    a.tiles = a.tiles_x * ceil_div(ceil_div(H, 2), TTH);
    const size_t lds = (size_t)Cfg::LDS_FLOATS * sizeof(float);
    static_assert(Cfg::LDS_FLOATS * sizeof(float) <= 160 * 1024, "LDS budget");
    static AttrMask attr_done = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(conv_wino_kernel<CIN, COUT, CB, TBG, NCBW, NTBW, TTH, TTW, COUT2, NHWC>), 160 * 1024, attr_done);
    int grid = xcd_grid_size(a.tiles, B);
    conv_wino_kernel<CIN, COUT, CB, TBG, NCBW, NTBW, TTH, TTW, COUT2, NHWC><<<grid, Cfg::NTHR, lds, st>>>(a);
    return 0;
}

static long wino_groups(int H, int W, int tth, int ttw) { return (long)ceil_div(ceil_div(H, 2), tth) * ceil_div(ceil_div(W, 2), ttw); }

int launch_conv_wino(const ConvW& c, const float* zeros, const float* in, int B, int H, int W, float* out, hipStream_t st, int cfg,
                     long long* trace, const ConvW* c2, bool nhwc) {
    (void)zeros;
    if (c.ks != 3 || c.stride != 1 || !c.w_wino) return -1;
    const int key = c.cin * 1000 + c.cout;
    if (c2) {      // 3x3 + fused 1x1
        if (c2->ks != 1 || c2->cin != c.cout || c2->cout != 64) return -1;
        const bool tall = wino_groups(H, W, 8, 4) < wino_groups(H, W, 4, 8);
        if (key == 64 * 1000 + 64) {
            if (nhwc) return tall ? run_wino<64, 64, 2, 1, 2, 1, 8, 4, 64, true>(c, in, B, H, W, out, st, trace, c2)
                                  : run_wino<64, 64, 2, 1, 2, 1, 4, 8, 64, true>(c, in, B, H, W, out, st, trace, c2);
            return tall ? run_wino<64, 64, 2, 1, 2, 1, 8, 4, 64, false>(c, in, B, H, W, out, st, trace, c2)
                        : run_wino<64, 64, 2, 1, 2, 1, 4, 8, 64, false>(c, in, B, H, W, out, st, trace, c2);
        }
        if (key == 128 * 1000 + 128 && !nhwc)
            return tall ? run_wino<128, 128, 4, 1, 2, 1, 8, 4, 64, false>(c, in, B, H, W, out, st, trace, c2)
                        : run_wino<128, 128, 4, 1, 2, 1, 4, 8, 64, false>(c, in, B, H, W, out, st, trace, c2);
        return -1;
    }
    if (nhwc) return -1;
    // cfg 0 = the production choice; cfg >= 1 = explicit variants (xfh_conv_layer variant 2, 3, ... for tuning)
    switch (key) {
        case 24 * 1000 + 24:     // 1 cout block: waves hold 2 tile blocks each
            if (cfg == 2) return run_wino<24, 24, 1, 2, 1, 2, 8, 8>(c, in, B, H, W, out, st, trace);   // 4-wave workgroups, 64 tiles, two per CU
            return run_wino<24, 24, 1, 4, 1, 2, 8, 16>(c, in, B, H, W, out, st, trace);                      // 8 waves, 128 tiles
        case 64 * 1000 + 64:     // waves hold both cout blocks of one tile block
            if (cfg == 2) return run_wino<64, 64, 2, 2, 2, 1, 8, 8>(c, in, B, H, W, out, st, trace);        // 8 waves, 64 tiles
            // (a persistent 8-wave variant -- outputs of tile k parked in LDS and stored under the MFMAs of tile k+1 -- measured -11 % back to back
            // on a hot cache but -0.5 ... -2 % inside the real step, needed 256 VGPRs + 28 B of scratch, and was removed: DESIGN 3.2)
            if (cfg == 3) return run_wino<64, 64, 2, 2, 2, 1, 16, 4>(c, in, B, H, W, out, st, trace);
            if (cfg == 4 || (cfg == 0 && wino_groups(H, W, 8, 4) < wino_groups(H, W, 4, 8)))
                return run_wino<64, 64, 2, 1, 2, 1, 8, 4>(c, in, B, H, W, out, st, trace);
            return run_wino<64, 64, 2, 1, 2, 1, 4, 8>(c, in, B, H, W, out, st, trace);                       // 4-wave workgroups, 32 tiles, two per CU
        case 128 * 1000 + 128:   // 4 cout blocks x 1 tile block: 32 tiles
            if (cfg == 2 || (cfg == 0 && wino_groups(H, W, 8, 4) < wino_groups(H, W, 4, 8)))
                return run_wino<128, 128, 4, 1, 2, 1, 8, 4>(c, in, B, H, W, out, st, trace);
            return run_wino<128, 128, 4, 1, 2, 1, 4, 8>(c, in, B, H, W, out, st, trace);
    }
    return -1;
}

}  // namespace xfh
