// Fused network heads on the f32 matrix cores (persistent kernels).
//
//   keypoint head  (modules/model.py:87-92,152 + modules/xfeat.py:242-247):
//       8x8 unfold of the normalised gray image -> 3 x [1x1 conv 64->64 + BN + ReLU] ->
//       1x1 conv 64->65 + bias -> softmax over the 65 logits -> drop dustbin ->
//       depth-to-space 8x8 -> heat (B,H,W)            [optionally also the raw logits]
//   reliability head (modules/model.py:79-84):
//       feats (channels-last) -> 2 x [1x1 conv 64->64 + BN + ReLU] -> 1x1 conv 64->1 -> sigmoid
//
// One unit of work = a tile of 256 cells (one workgroup of 8 waves, 32 cells per wave).
// Orientation as in k_conv_mfma.hip: D[feature][cell] (A = weights, B = activations), so a
// layer's ReLU'd accumulator registers ARE the next layer's B operand when the K pairing is
// chosen as channels (c, c+4): the whole chain lives in registers.  Only the first layer reads
// activations from LDS ([cell][65] floats, conflict-free), filled by the global->LDS DMA
// (one 64-lane dword copy per cell: the 8x8 unfold is just the DMA's source addressing).
// The kernels are persistent: grid = #CUs, the head's weights are copied to LDS once per
// workgroup, and the next tile's activations are in flight while layers 2..4 run.
#include "kernels.hpp"
#include "bx_split.hpp"
#include "head_bx_body.hpp"
#include "head_f32r_body.hpp"
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace xfh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int HD_XS = 65;        // LDS row stride of the activation tile

// SHIFT (debug, tools/head_soak.py; torture build only): the kernel body moved by SHIFT x 4 bytes against the 64-byte instruction-cache lines
template <bool KP, int SHIFT = 0>
__global__ __launch_bounds__(512) void head_fused_kernel(HeadArgs a) {
    code_shift<SHIFT>();
    kernel_entry_hooks(a.cold);      // debug: code-position shift / cold instruction cache (common.hpp)
    constexpr int NL = KP ? 4 : 3;
    constexpr int W_FLOATS = KP ? (3 * 64 * 64 + 64 * 96) : (2 * 64 * 64 + 64);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;
    float* Xl = smem + W_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = a.hc * a.wc;

    // all weights of this head -> LDS, once
    {
        int off = 0;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            const int n = (KP && l == 3) ? 64 * 96 : ((!KP && l == 2) ? 64 : 64 * 64);
            if (n >= 256) {
                for (int j = wave; j < n / 256; j += 8)
                    __builtin_amdgcn_global_load_lds((gptr_t)(a.w[l] + j * 256 + lane * 4), (lptr_t)(Wl + off + j * 256), 16, 0, 0);
            } else if (wave == 0) {
                __builtin_amdgcn_global_load_lds((gptr_t)(a.w[l] + lane), (lptr_t)(Wl + off), 4, 0, 0);
            }
            off += n;
        }
    }
    // activation tile of `tile` -> LDS [cell][65]: one 64-lane dword DMA per cell
    auto issue_x = [&](int tile) {
#pragma unroll 4
        for (int cc = 0; cc < 32; ++cc) {
            const int cl = wave * 32 + cc;
            const int g = tile * HD_CELLS + cl;             // wave-uniform
            const float* p = a.zeros + lane;
            if (g < a.ncell) {
                if (KP) {
                    const int b = g / hw, rem = g - b * hw;
                    const int ci = rem / a.wc, cj = rem - ci * a.wc;
                    p = a.src + (size_t)b * a.H * a.W + (size_t)(8 * ci + (lane >> 3)) * a.W + 8 * cj + (lane & 7);
                } else {
                    p = a.src + (size_t)g * 64 + lane;
                }
            }
            __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(Xl + cl * HD_XS), 4, 0, 0);
        }
    };

    int tile = blockIdx.x;
    if (tile < a.ntiles) issue_x(tile);
    for (; tile < a.ntiles; tile += gridDim.x) {
        lds_dma_barrier();                                 // tile (and the weights) landed
        // ---- layer 1: B operand from LDS ------------------------------------------------------
        f32x16 accA[2], accB[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) accA[m][r] = a.bias[0][m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
        {
            const float* xb = Xl + (wave * 32 + l31) * HD_XS + half;
            const float* wb = Wl + half * 64 + l31;
            // key-point head: the tile holds RAW gray; this lane's cell belongs to image cb -> x = fmaf(g, alpha, beta)
            float nalpha = 1.f, nbeta = 0.f;
            if (KP) {
                const int cg = min(tile * HD_CELLS + wave * 32 + l31, a.ncell - 1);
                const int cb = cg / hw;
                nalpha = a.coef[2 * cb]; nbeta = a.coef[2 * cb + 1];
            }
            float av[2][2], bv[2];
            auto ld = [&](int p, float (&ao)[2], float& bo) {
                ao[0] = wb[(2 * p) * 64];
                ao[1] = wb[(2 * p) * 64 + 32];
                bo = xb[2 * p];
            };
            ld(0, av[0], bv[0]);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            float nrm2 = 0.f;          // REL: this lane walks 32 of its cell's 64 channels anyway -> squared norm for free
#pragma unroll
            for (int p = 0; p < 32; ++p) {
                if (p + 1 < 32) ld(p + 1, av[(p + 1) & 1], bv[(p + 1) & 1]);
                const float xv = KP ? fmaf(bv[p & 1], nalpha, nbeta) : bv[p & 1];
                if (!KP) nrm2 = fmaf(xv, xv, nrm2);
                accA[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p & 1][0], xv, accA[0], 0, 0, 0);
                accA[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p & 1][1], xv, accA[1], 0, 0, 0);
                if (p + 1 < 32) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
            }
            if (!KP && a.inv) {
                nrm2 += xhalf(nrm2);                       // the other 32 channels sit in the other half-wave
                const int gc = tile * HD_CELLS + wave * 32 + l31;
                if (half == 0 && gc < a.ncell) a.inv[gc] = 1.f / fmaxf(sqrtf(nrm2), 1e-12f);
            }
        }
        lds_dma_barrier();                                 // every wave is done with the X tile
        if (tile + (int)gridDim.x < a.ntiles) issue_x(tile + gridDim.x);   // next tile flies during layers 2..

        const int gcell = tile * HD_CELLS + wave * 32 + l31;            // this lane's cell
        if (KP) {
            chain_layer<2>(Wl + 64 * 64, 64, a.bias[1], accA, accB, l31, half);
            chain_layer<2>(Wl + 2 * 64 * 64, 64, a.bias[2], accB, accA, l31, half);
            f32x16 lg[3];
            chain_layer<3>(Wl + 3 * 64 * 64, 96, a.bias[3], accA, lg, l31, half);
            // lane (l31,half) holds logits c = 32m + (r&3) + 8(r>>2) + 4*half of its cell; c == 64 (dustbin) is m=2,r=0,half=0
            float mx = -INFINITY;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, lg[m][r]);
            if (half == 0) mx = fmaxf(mx, lg[2][0]);
            mx = fmaxf(mx, xhalf(mx));
            float sum = 0.f;
            f32x16 e[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) { e[m][r] = expf(lg[m][r] - mx); sum += e[m][r]; }
            if (half == 0) sum += expf(lg[2][0] - mx);
            sum += xhalf(sum);
            if (gcell < a.ncell) {
                const int b = gcell / hw, rem = gcell - b * hw;
                const int ci = rem / a.wc, cj = rem - ci * a.wc;
                float* o = a.out + (size_t)b * a.H * a.W + (size_t)(8 * ci) * a.W + 8 * cj + 4 * half;
                const float rs = 1.f / sum;                // one correctly-rounded divide, then 64 multiplies
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {          // dy = q + 4m, dx = 4*half .. +3
                        const float4 v = make_float4(e[m][4 * q] * rs, e[m][4 * q + 1] * rs, e[m][4 * q + 2] * rs, e[m][4 * q + 3] * rs);
                        *reinterpret_cast<float4*>(o + (size_t)(q + 4 * m) * a.W) = v;
                    }
                if (a.logits) {
                    float* lp = a.logits + (size_t)gcell * 65;
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int r = 0; r < 16; ++r) lp[m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] = lg[m][r];
                    if (half == 0) lp[64] = lg[2][0];
                }
            }
        } else {
            chain_layer<2>(Wl + 64 * 64, 64, a.bias[1], accA, accB, l31, half);
            // final 64 -> 1: dot over this lane's 32 channels, other half via one shuffle
            const float* w3 = Wl + 2 * 64 * 64;
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    s = fmaf(fmaxf(accB[m][r], 0.f), w3[m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half], s);
            s += __shfl_xor(s, 32, 64);
            if (half == 0 && gcell < a.ncell) a.out[gcell] = 1.f / (1.f + expf(-(s + a.bias[2][0])));
        }
    }
}


// the default heads: body in head_f32r_body.hpp (also compiled for the host by tests/emu/)
template <bool KP, int SHIFT = 0, bool DUST = true>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void head_f32r_kernel(HeadArgs a) {
    code_shift<SHIFT>();
    kernel_entry_hooks(a.cold);      // debug: code-position shift / cold instruction cache (common.hpp)
    head_f32r_body<KP, DUST>(a);
}

// the split-operand heads: body in head_bx_body.hpp (also compiled for the host by tests/emu/)
template <bool KP, int SHIFT = 0, int FXM = 0>      // FXM: 0 bf16 three-way split, 1 .. 3 the fp16-pair forms (head_bx_body.hpp)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void head_bx_kernel(HeadBxArgs a) {
    code_shift<SHIFT>();
    kernel_entry_hooks(a.cold);      // debug: code-position shift / cold instruction cache (common.hpp)
    head_bx_body<KP, FXM>(a);
}
// dynamic LDS of a form: weight fragments (1 KiB each), biases, (FXM 3) a 4-KiB slot pair per wave
static size_t head_bx_lds(bool kp, int fxm) { return (size_t)(kp ? (fxm ? 8 : 9) : 4) * 4 * (fxm == 2 ? 2 : 3) * 1024 + ((kp ? 288 : 128) + 64) * sizeof(float) + (fxm == 3 ? 8 * 4096 : 0); }
template <bool KP, int SHIFT, int FXM>
static void launch_head_bx(const HeadBxArgs& h, hipStream_t st) {
    static AttrMask attr = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(head_bx_kernel<KP, SHIFT, FXM>), 160 * 1024, attr);
    head_bx_kernel<KP, SHIFT, FXM><<<min(h.ntiles, num_cus()), 512, head_bx_lds(KP, FXM), st>>>(h);
}
// option fx -> the form: bit 8 the fp16 pair, + 16 two weight fragments in LDS, + 32 (instead) B through LDS
static int head_fxm(int fx, const NetWeights& nw, int hd) {
    if (!(fx & 8) || !nw.head_fx[hd]) return 0;
    return (fx & 32) ? 3 : (fx & 16) ? 2 : 1;
}

long long* g_head_trace = nullptr;        // debug (xfh_debug_trace): stamps of head_bx_kernel<true>

void launch_kp_head(const NetWeights& nw, const float* gray, const float* coef, int B, int H, int W, float* heat, float* logits, hipStream_t st, int f32_kernels, int fx, int* status) {
    if (!f32_kernels && nw.head_bx[0]) {
        const int fxm = head_fxm(fx, nw, 0);
        HeadBxArgs h{};
        h.cold = g_debug_cold;
        h.status = status;
        h.src = gray; h.coef = coef; h.wq = reinterpret_cast<const uint4*>(fxm == 2 ? nw.head_fq[0] : fxm ? nw.head_fx[0] : nw.head_bx[0]); h.bias = nw.head_bx_bias[0]; h.out = heat; h.logits = logits;
        h.H = H; h.W = W; h.hc = H / 8; h.wc = W / 8;
        h.ncell = B * h.hc * h.wc;
        h.ntiles = ceil_div(h.ncell, HD_CELLS);
        h.trace = g_head_trace;
        h.w_dust = nw.conv[L_KP_3].w_oihw + 64 * 64; h.b_dust = nw.head_kp_b_dust;
        if (fxm == 3) launch_head_bx<true, 0, 3>(h, st);
        else if (fxm == 2) launch_head_bx<true, 0, 2>(h, st);
        else if (fxm == 1) launch_head_bx<true, 0, 1>(h, st);
        else launch_head_bx<true, 0, 0>(h, st);
        return;
    }
    HeadArgs a{};
    a.cold = g_debug_cold;
    a.src = gray; a.coef = coef; a.zeros = nw.zeros; a.out = heat; a.logits = logits;
    a.H = H; a.W = W; a.hc = H / 8; a.wc = W / 8;
    a.ncell = B * a.hc * a.wc;
    a.ntiles = ceil_div(a.ncell, HD_CELLS);
    const int L[4] = {L_KP_0, L_KP_1, L_KP_2, L_KP_3};
    for (int i = 0; i < 4; ++i) { a.w[i] = nw.conv[L[i]].w_kcp; a.bias[i] = nw.conv[L[i]].bias; }
    if (f32_kernels >= 2) {      // the register-input form (no activation tile, no barrier per tile); 3: with the dustbin logit on the matrix cores (round 4)
        static AttrMask attr_r{0}, attr_o{0};
        if (f32_kernels == 3) {
            set_max_dynamic_lds(reinterpret_cast<const void*>(head_f32r_kernel<true, 0, false>), 160 * 1024, attr_o);
            head_f32r_kernel<true, 0, false><<<min(a.ntiles, num_cus()), 512, (size_t)(3 * 64 * 64 + 64 * 96) * sizeof(float), st>>>(a);
            return;
        }
        set_max_dynamic_lds(reinterpret_cast<const void*>(head_f32r_kernel<true>), 160 * 1024, attr_r);
        head_f32r_kernel<true><<<min(a.ntiles, num_cus()), 512, (size_t)(3 * 64 * 64 + 64 * 96) * sizeof(float), st>>>(a);
        return;
    }
    const size_t lds = (size_t)(3 * 64 * 64 + 64 * 96 + HD_CELLS * HD_XS) * sizeof(float);
    static AttrMask attr = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(head_fused_kernel<true>), 160 * 1024, attr);
    head_fused_kernel<true><<<min(a.ntiles, num_cus()), 512, lds, st>>>(a);
}

void launch_rel_head(const NetWeights& nw, const float* feats, int ncell, float* reliab, float* invnorm, hipStream_t st, int f32_kernels, int fx, int* status) {
    if (!f32_kernels && nw.head_bx[1]) {
        const int fxm = head_fxm(fx, nw, 1);
        HeadBxArgs h{};
        h.cold = g_debug_cold;
        h.status = status;
        h.src = feats; h.wq = reinterpret_cast<const uint4*>(fxm == 2 ? nw.head_fq[1] : fxm ? nw.head_fx[1] : nw.head_bx[1]); h.bias = nw.head_bx_bias[1]; h.out = reliab; h.inv = invnorm;
        h.w_last = nw.conv[L_HEAT_2].w_oihw; h.b_last = nw.head_rel_b_last;
        h.hc = 1; h.wc = 1; h.H = 8; h.W = 8;
        h.ncell = ncell;
        h.ntiles = ceil_div(ncell, HD_CELLS);
        if (fxm == 3) launch_head_bx<false, 0, 3>(h, st);
        else if (fxm == 2) launch_head_bx<false, 0, 2>(h, st);
        else if (fxm == 1) launch_head_bx<false, 0, 1>(h, st);
        else launch_head_bx<false, 0, 0>(h, st);
        return;
    }
    HeadArgs a{};
    a.cold = g_debug_cold;
    a.src = feats; a.zeros = nw.zeros; a.out = reliab; a.logits = nullptr; a.inv = invnorm;
    a.hc = 1; a.wc = 1; a.H = 8; a.W = 8;
    a.ncell = ncell;
    a.ntiles = ceil_div(ncell, HD_CELLS);
    a.w[0] = nw.conv[L_HEAT_0].w_kcp; a.bias[0] = nw.conv[L_HEAT_0].bias;
    a.w[1] = nw.conv[L_HEAT_1].w_kcp; a.bias[1] = nw.conv[L_HEAT_1].bias;
    a.w[2] = nw.conv[L_HEAT_2].w_oihw; a.bias[2] = nw.conv[L_HEAT_2].bias;
    if (f32_kernels >= 2) {
        static AttrMask attr_r = 0;
        set_max_dynamic_lds(reinterpret_cast<const void*>(head_f32r_kernel<false>), 160 * 1024, attr_r);
        head_f32r_kernel<false><<<min(a.ntiles, num_cus()), 512, (size_t)(2 * 64 * 64 + 64) * sizeof(float), st>>>(a);
        return;
    }
    const size_t lds = (size_t)(2 * 64 * 64 + 64 + HD_CELLS * HD_XS) * sizeof(float);
    static AttrMask attr = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(head_fused_kernel<false>), 160 * 1024, attr);
    head_fused_kernel<false><<<min(a.ntiles, num_cus()), 512, lds, st>>>(a);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Debug (xfh_debug_head_soak, tools/head_soak.py): the key-point head alone, launched `iters` times, every result compared on the
// device with a reference result; differing float4s are counted and the first `cap` of them recorded as {iteration, float4 index,
// bits got, bits expected} behind a 4-word header {count, 0, 0, 0}.  variant: 0 = the split-bf16 kernel, 100 = the f32-MFMA kernel with
// an activation tile (head_fused_kernel), 101 = the f32-MFMA kernel with register input (head_f32r_kernel, the default head), 102 / 103 / 104 = the split head in the fp16-pair arithmetic (three weight fragments | two | three and B through LDS);
// 1000 + s / 2000 + s / 3000 + s / 4000 + s / 5000 + s = 0, 100, 101, 102, 104 COLD-STARTED (s_icache_inv per workgroup) with the body moved by 4 s bytes,
// s = 0 .. 15: the code-position scan that separates a kernel that trips on instruction-cache refills from one that does not.
// (The experiment builds of round 4 -- reloads, pads, dumps, dry passes: variants 1 .. 26 of profiles/r04_head_hazard -- lived here until commit 9607d16.)
// ------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void soak_compare_kernel(const uint4* __restrict__ got, const uint4* __restrict__ ref, size_t n4, int iter, unsigned* rep, unsigned cap) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const uint4 g = got[i], r = ref[i];
        if (g.x != r.x || g.y != r.y || g.z != r.z || g.w != r.w) {
            const unsigned k = atomicAdd(rep, 1u);
            if (k < cap) {
                unsigned* o = rep + 4 + 4 * (size_t)k;
                const int c = g.x != r.x ? 0 : g.y != r.y ? 1 : g.z != r.z ? 2 : 3;
                o[0] = (unsigned)iter; o[1] = (unsigned)i; o[2] = (&g.x)[c]; o[3] = (&r.x)[c];
            }
        }
    }
}

template <int SHIFT>
static void launch_kp_head_bx_shift(const HeadBxArgs& h, hipStream_t st) { launch_head_bx<true, SHIFT, 0>(h, st); }
template <int SHIFT>
static void launch_kp_head_fx_shift(const HeadBxArgs& h, hipStream_t st) { launch_head_bx<true, SHIFT, 1>(h, st); }
template <int SHIFT>
static void launch_kp_head_fl_shift(const HeadBxArgs& h, hipStream_t st) { launch_head_bx<true, SHIFT, 3>(h, st); }
template <int SHIFT>
static void launch_kp_head_f32_shift(const HeadArgs& a, hipStream_t st) {
    const size_t lds = (size_t)(3 * 64 * 64 + 64 * 96 + HD_CELLS * HD_XS) * sizeof(float);
    static AttrMask attr = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(head_fused_kernel<true, SHIFT>), 160 * 1024, attr);
    head_fused_kernel<true, SHIFT><<<min(a.ntiles, num_cus()), 512, lds, st>>>(a);
}
template <int SHIFT>
static void launch_kp_head_f32r_shift(const HeadArgs& a, hipStream_t st) {
    static AttrMask attr = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(head_f32r_kernel<true, SHIFT>), 160 * 1024, attr);
    head_f32r_kernel<true, SHIFT><<<min(a.ntiles, num_cus()), 512, (size_t)(3 * 64 * 64 + 64 * 96) * sizeof(float), st>>>(a);
}
// The scan's 3 x 16 instantiations are compiled into the torture build only (python -m accelerated_features_amd.build --scan -> libxfeat_hip_scan.so, defines
// XFH_HEAD_SCAN_SHIFTS = 16); the production library holds position 0 of each kernel.
#ifndef XFH_HEAD_SCAN_SHIFTS
#define XFH_HEAD_SCAN_SHIFTS 1
#endif
template <int S>
static bool launch_shift(int shift, const HeadBxArgs& h, const HeadBxArgs& hx, const HeadBxArgs& hq, const HeadArgs& a, int kind, hipStream_t st) {      // shift -> the instantiation; kind 1 bf16, 2 f32 (LDS tile), 3 f32 (registers), 4 fp16 pair, 5 fp16 pair with B through LDS
    if (shift == S) {
        if (kind == 2) launch_kp_head_f32_shift<S>(a, st); else if (kind == 3) launch_kp_head_f32r_shift<S>(a, st); else if (kind == 4) launch_kp_head_fx_shift<S>(hx, st);
        else if (kind == 5) launch_kp_head_fl_shift<S>(hx, st);
        else launch_kp_head_bx_shift<S>(h, st);
        return true;
    }
    if constexpr (S + 1 < XFH_HEAD_SCAN_SHIFTS) return launch_shift<S + 1>(shift, h, hx, hq, a, kind, st);
    return false;
}

int head_soak(const NetWeights& nw, const float* gray, const float* coef, int B, int H, int W, float* heat, const float* heat_ref, float* logits, const float* logits_ref,
              int variant, int iters, int iter0, unsigned* rep_heat, unsigned* rep_logits, unsigned cap, hipStream_t st) {
    HeadBxArgs h{};
    h.src = gray; h.coef = coef; h.wq = reinterpret_cast<const uint4*>(nw.head_bx[0]); h.bias = nw.head_bx_bias[0]; h.out = heat; h.logits = logits;
    h.H = H; h.W = W; h.hc = H / 8; h.wc = W / 8;
    h.ncell = B * h.hc * h.wc;
    h.ntiles = ceil_div(h.ncell, HD_CELLS);
    HeadArgs fa{};
    fa.src = gray; fa.coef = coef; fa.zeros = nw.zeros; fa.out = heat; fa.logits = logits; fa.H = H; fa.W = W; fa.hc = H / 8; fa.wc = W / 8; fa.ncell = h.ncell; fa.ntiles = h.ntiles;
    {
        const int L[4] = {L_KP_0, L_KP_1, L_KP_2, L_KP_3};
        for (int i = 0; i < 4; ++i) { fa.w[i] = nw.conv[L[i]].w_kcp; fa.bias[i] = nw.conv[L[i]].bias; }
    }
    fa.cold = h.cold = (variant >= 1000) ? 1 : g_debug_cold;
    h.w_dust = nw.conv[L_KP_3].w_oihw + 64 * 64; h.b_dust = nw.head_kp_b_dust;
    HeadBxArgs hx = h;                     // the fp16-pair head (variants 102, 4000 + s)
    hx.wq = reinterpret_cast<const uint4*>(nw.head_fx[0]);
    HeadBxArgs hq = h;                     // its two-fragment image (variant 103)
    hq.wq = reinterpret_cast<const uint4*>(nw.head_fq[0]);
    const size_t n4h = (size_t)B * H * W / 4, n4l = (size_t)h.ncell * 65 / 4;
    for (int it = 0; it < iters; ++it) {
        if (variant >= 1000) {
            if (variant >= 6000 || (variant >= 4000 && !nw.head_fx[0]) || !launch_shift<0>(variant % 1000, h, hx, hq, fa, variant / 1000, st)) return -1;
        } else if (variant == 0) launch_kp_head_bx_shift<0>(h, st);
        else if (variant == 102 && nw.head_fx[0]) launch_kp_head_fx_shift<0>(hx, st);
        else if (variant == 103 && nw.head_fx[0]) launch_head_bx<true, 0, 2>(hq, st);
        else if (variant == 104 && nw.head_fx[0]) launch_kp_head_fl_shift<0>(hx, st);
        else if (variant == 100) launch_kp_head_f32_shift<0>(fa, st);
        else if (variant == 101) launch_kp_head_f32r_shift<0>(fa, st);
        else return -1;
        if (heat_ref) soak_compare_kernel<<<1024, 256, 0, st>>>(reinterpret_cast<const uint4*>(heat), reinterpret_cast<const uint4*>(heat_ref), n4h, iter0 + it, rep_heat, cap);
        if (logits && logits_ref) soak_compare_kernel<<<1024, 256, 0, st>>>(reinterpret_cast<const uint4*>(logits), reinterpret_cast<const uint4*>(logits_ref), n4l, iter0 + it, rep_logits, cap);
    }
    return 0;
}

}  // namespace xfh
