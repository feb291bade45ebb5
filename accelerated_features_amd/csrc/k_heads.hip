// Fused network heads (persistent kernels).
//
//   keypoint head  (modules/model.py:87-92,152 + modules/xfeat.py:242-247):
//       8x8 unfold of the normalised gray image -> 3 x [1x1 conv 64->64 + BN + ReLU] ->
//       1x1 conv 64->65 + bias -> softmax over the 65 logits -> drop dustbin ->
//       depth-to-space 8x8 -> heat (B,H,W)            [optionally also the raw logits]
//   reliability head (modules/model.py:79-84):
//       feats (channels-last) -> 2 x [1x1 conv 64->64 + BN + ReLU] -> 1x1 conv 64->1 -> sigmoid
//
// One unit of work = a tile of 256 cells (one workgroup of 8 waves, 32 cells per wave).  Orientation: D[feature][cell] (A = weights, B = activations), so a
// layer's ReLU'd accumulator registers ARE the next layer's B operand: the whole chain lives in registers.  The kernels are persistent: grid = #CUs, the head's
// weights are copied to LDS once per workgroup, and the next tile's first-layer input is in flight while the chained layers run.
//   head_bx_kernel   (head_bx_body.hpp):   the default -- fp16 matrix cores, fp16-pair arithmetic (fp32-equivalent; range-guarded)
//   head_f32r_kernel (head_f32r_body.hpp): the same chain on v_mfma_f32_32x32x2_f32 -- fp32's range: the fallback when an activation or a weight leaves the pair's range
#include "kernels.hpp"
#include "bx_split.hpp"
#include "head_bx_body.hpp"
#include "head_f32r_body.hpp"

namespace xfh {

template <bool KP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void head_f32r_kernel(HeadArgs a) {
    kernel_entry_hooks(a.cold);      // debug: code-position shift / cold instruction cache (common.hpp)
    head_f32r_body<KP, true>(a);
}

template <bool KP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void head_bx_kernel(HeadBxArgs a) {
    kernel_entry_hooks(a.cold);      // debug: code-position shift / cold instruction cache (common.hpp)
    head_bx_body<KP>(a);
}
// dynamic LDS: weight fragments (1 KiB each), biases, the dustbin weights
static size_t head_bx_lds(bool kp) { return (size_t)(kp ? 8 : 4) * 4 * 3 * 1024 + ((kp ? 288 : 128) + 64) * sizeof(float); }
template <bool KP>
static void launch_head_bx(const HeadBxArgs& h, hipStream_t st) {
    static AttrMask attr = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(head_bx_kernel<KP>), 160 * 1024, attr);
    head_bx_kernel<KP><<<min(h.ntiles, num_cus()), 512, head_bx_lds(KP), st>>>(h);
}

long long* g_head_trace = nullptr;        // debug (xfh_debug_trace): stamps of head_bx_kernel<true>

// f32_kernels: run the f32-MFMA kernel.  The fp16-pair kernel is also left when the head has no fp16-pair weights (a |w| >= kFxMaxWeight: NetWeights::head_fx is NULL)
void launch_kp_head(const NetWeights& nw, const float* gray, const float* coef, int B, int H, int W, float* heat, float* logits, hipStream_t st, bool f32_kernels, int* status) {
    if (!f32_kernels && nw.head_fx[0]) {
        HeadBxArgs h{};
        h.cold = g_debug_cold;
        h.status = status;
        h.src = gray; h.coef = coef; h.wq = reinterpret_cast<const uint4*>(nw.head_fx[0]); h.bias = nw.head_bx_bias[0]; h.out = heat; h.logits = logits;
        h.H = H; h.W = W; h.hc = H / 8; h.wc = W / 8;
        h.ncell = B * h.hc * h.wc;
        h.ntiles = ceil_div(h.ncell, HD_CELLS);
        h.trace = g_head_trace;
        h.w_dust = nw.conv[L_KP_3].w_oihw + 64 * 64; h.b_dust = nw.head_kp_b_dust;
        launch_head_bx<true>(h, st);
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
    static AttrMask attr_r{0};
    set_max_dynamic_lds(reinterpret_cast<const void*>(head_f32r_kernel<true>), 160 * 1024, attr_r);
    head_f32r_kernel<true><<<min(a.ntiles, num_cus()), 512, (size_t)(3 * 64 * 64 + 64 * 96) * sizeof(float), st>>>(a);
}

void launch_rel_head(const NetWeights& nw, const float* feats, int ncell, float* reliab, float* invnorm, hipStream_t st, bool f32_kernels, int* status) {
    if (!f32_kernels && nw.head_fx[1]) {
        HeadBxArgs h{};
        h.cold = g_debug_cold;
        h.status = status;
        h.src = feats; h.wq = reinterpret_cast<const uint4*>(nw.head_fx[1]); h.bias = nw.head_bx_bias[1]; h.out = reliab; h.inv = invnorm;
        h.w_last = nw.conv[L_HEAT_2].w_oihw; h.b_last = nw.head_rel_b_last;
        h.hc = 1; h.wc = 1; h.H = 8; h.W = 8;
        h.ncell = ncell;
        h.ntiles = ceil_div(ncell, HD_CELLS);
        launch_head_bx<false>(h, st);
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
    static AttrMask attr_r{0};
    set_max_dynamic_lds(reinterpret_cast<const void*>(head_f32r_kernel<false>), 160 * 1024, attr_r);
    head_f32r_kernel<false><<<min(a.ntiles, num_cus()), 512, (size_t)(2 * 64 * 64 + 64) * sizeof(float), st>>>(a);
}

}  // namespace xfh
