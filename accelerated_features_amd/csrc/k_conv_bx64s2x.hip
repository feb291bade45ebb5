// 3x3 stride-2 convolution, 64 -> 64 or 64 -> 128 channels (block4.0 / block5.0; modules/model.py:68,75) in the fp16-pair arithmetic
// (body and notes: conv_bx64s2_body.hpp, also compiled for the host by tests/emu/).
#include "kernels.hpp"
#include "bx_split.hpp"
#include <type_traits>
#include "conv_bx64s2_body.hpp"

namespace xfh {

template <int NCO, bool W4>      // NCO: cout halves; W4: W % 4 == 0
__global__ __launch_bounds__(768) __attribute__((amdgpu_waves_per_eu(3, 3)))
void conv_bx64s2x_kernel(Bx64S2xArgs a) {
    kernel_entry_hooks(a.cold);      // debug: code-position shift / cold instruction cache (common.hpp)
    conv_bx64s2w_body<NCO, W4>(a);
}

template <int NCO, bool W4>
static int run_bx64s2x(const ConvW& c, const float* in, int B, int H, int W, float* out, hipStream_t st, long long* trace, int* status) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if ((size_t)64 * H * W * sizeof(float) >= 0x7fffffffu) return -1;      // buffer-resource range
    Bx64S2xArgs a;
    a.cold = g_debug_cold;
    a.status = status;
    a.in = in; a.wq = c.w_fx; a.bias = c.bias; a.out = out; a.relu = c.relu; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.B = B; a.trace = trace;
    a.nrows = ceil_div(Ho, 8); a.upi = ceil_div(Wo, 16) * a.nrows;
    static AttrMask attr_done = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(conv_bx64s2x_kernel<NCO, W4>), bx64s2x::LDS_W_BYTES, attr_done);
    const long long units = (long long)NCO * B * a.upi;
    int grid = num_cus();                      // one 8-wave workgroup per CU; a multiple of 8 keeps a workgroup on its XCD
    if (units < grid) grid = (int)units;
    conv_bx64s2x_kernel<NCO, W4><<<grid, 768, bx64s2x::LDS_W_BYTES, st>>>(a);
    return 0;
}

// -1: no fp16-pair image of the layer's weights (a weight beyond the fp16 range), or not this kernel's layer
int launch_conv_bx64s2_fx(const ConvW& c, const float* in, int B, int H, int W, float* out, hipStream_t st, long long* trace, int* status) {
    if (c.ks != 3 || c.stride != 2 || !c.w_fx || c.cin != 64) return -1;
    const bool w4 = (W & 3) == 0;
    if (c.cout == 64) return w4 ? run_bx64s2x<1, true>(c, in, B, H, W, out, st, trace, status) : run_bx64s2x<1, false>(c, in, B, H, W, out, st, trace, status);
    if (c.cout == 128) return w4 ? run_bx64s2x<2, true>(c, in, B, H, W, out, st, trace, status) : run_bx64s2x<2, false>(c, in, B, H, W, out, st, trace, status);
    return -1;
}

}  // namespace xfh
