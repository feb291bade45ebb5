// 3x3 stride-1 convolution, 64 -> 64 channels, fp16-pair arithmetic, weights resident in registers (K split over the four waves of a workgroup, partial sums
// reduced through LDS, padded-raster walk of the map): body and design notes in conv_rs64_body.hpp (also compiled for the host by tests/emu/).
#include "kernels.hpp"
#include "conv_rs64_body.hpp"

namespace xfh {

template <int FUSE, int CIN = 64, bool TRACE = false>      // FUSE 0: the 3x3 alone; 1: + trailing 1x1, NCHW output; 2: the same, channels-last output.  CIN 128: the 128 -> 128 layers.  TRACE: with s_memtime stamps (xfh_debug_trace)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void conv_rs64_kernel(Rs64Args a) {
    kernel_entry_hooks(a.cold);      // debug: code-position shift / cold instruction cache (common.hpp)
    conv_rs64_body<FUSE, CIN, TRACE>(a);
}

template <int FUSE>
static int run_rs64(const ConvW& c, const ConvW* c2, const float* in, int B, int H, int W, float* out, hipStream_t st, int* status, long long* trace) {
    if ((size_t)64 * H * W * sizeof(float) >= 0x7fffffffu) return -1;      // buffer-resource range
    Rs64Args a;
    a.cold = g_debug_cold;
    a.status = status;
    a.in = in; a.wq = c.w_rs; a.bias = c.bias; a.out = out; a.relu = c.relu; a.H = H; a.W = W; a.B = B;
    a.wq2 = c2 ? c2->w_rs : nullptr; a.bias2 = c2 ? c2->bias : nullptr; a.relu2 = c2 ? c2->relu : 0;
    a.trace = trace;
    rs64::strips_for(W, rs64::max_nseg(FUSE != 0), a.ns, a.ws);              // maps wider than the rings reach (125 columns; 93 with the fused 1x1's buffers) run as column strips
    a.P = a.ws + 2; a.inv_p = 1.f / (float)a.P; a.nu = ceil_div(H * a.P, 64); a.nseg = rs64::nseg_for(a.P);
    if ((long long)(H + 4) * a.P + 512 >= (1 << 20)) return -1;               // row_of (conv_rs64_body.hpp): positions below 2^20 (exact there for every P <= 127: tests/test_conv_rs64_emulated.py)
    const int lds = rs64::lds_bytes(a.nseg, FUSE != 0);
    static AttrMask attr_done = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(conv_rs64_kernel<FUSE>), rs64::lds_bytes(rs64::max_nseg(FUSE != 0), FUSE != 0), attr_done);
    int grid = num_cus();                      // one workgroup (four waves, one per SIMD) per CU
    a.k = rs64::runs_per_image(B * a.ns, a.nu, grid);
    const long long nruns = (long long)B * a.ns * a.k;
    if (nruns < grid) grid = (int)nruns;
    if constexpr (FUSE != 1) {
        if (trace) {      // the stamped twin (debug: xfh_debug_trace)
            static AttrMask attr_done_t = 0;
            set_max_dynamic_lds(reinterpret_cast<const void*>(conv_rs64_kernel<FUSE, 64, true>), rs64::lds_bytes(rs64::max_nseg(FUSE != 0), FUSE != 0), attr_done_t);
            conv_rs64_kernel<FUSE, 64, true><<<grid, 256, lds, st>>>(a);
            return 0;
        }
    }
    conv_rs64_kernel<FUSE><<<grid, 256, lds, st>>>(a);
    return 0;
}

// -1: not this kernel's layer or map (the caller keeps conv_bx64_kernel)
int launch_conv_rs64(const ConvW& c, const float* in, int B, int H, int W, float* out, hipStream_t st, int* status, const ConvW* c2, bool nhwc, long long* trace) {
    if (c.ks != 3 || c.stride != 1 || c.cout != 64 || c.cin != 64 || !c.w_rs) return -1;
    if (c2 && (c2->ks != 1 || c2->cin != 64 || c2->cout != 64 || !c2->w_rs)) return -1;
    if (!c2) return nhwc ? -1 : run_rs64<0>(c, nullptr, in, B, H, W, out, st, status, trace);
    return nhwc ? run_rs64<2>(c, c2, in, B, H, W, out, st, status, trace) : run_rs64<1>(c, c2, in, B, H, W, out, st, status, trace);
}

bool conv_rs128_fits(int W) { (void)W; return true; }      // (any width since the column strips: kept for the backbone's layer plan)

// 128 -> 128 3x3/s1 (block5.1, block5.2): four workgroups per run, one per cout quarter.  -1: not this kernel's layer, or the map is wider than 61 columns
int launch_conv_rs128(const ConvW& c, const float* in, int B, int H, int W, float* out, hipStream_t st, int* status) {
    if (c.ks != 3 || c.stride != 1 || c.cout != 128 || c.cin != 128 || !c.w_rs) return -1;
    if ((size_t)128 * H * W * sizeof(float) >= 0x7fffffffu) return -1;      // buffer-resource range
    Rs64Args a;
    a.cold = g_debug_cold;
    a.status = status;
    a.in = in; a.wq = c.w_rs; a.bias = c.bias; a.out = out; a.relu = c.relu; a.H = H; a.W = W; a.B = B;
    a.wq2 = nullptr; a.bias2 = nullptr; a.relu2 = 0; a.trace = nullptr;
    rs64::strips_for(W, rs64::MAX_NSEG128, a.ns, a.ws);                      // (maps wider than 61 columns: column strips)
    a.P = a.ws + 2; a.inv_p = 1.f / (float)a.P; a.nu = ceil_div(H * a.P, 64); a.nseg = rs64::nseg_for(a.P);
    if ((long long)(H + 4) * a.P + 512 >= (1 << 20)) return -1;
    static AttrMask attr_done = 0;
    set_max_dynamic_lds(reinterpret_cast<const void*>(conv_rs64_kernel<0, 128>), rs64::lds_bytes128(rs64::MAX_NSEG128), attr_done);
    int groups = num_cus() / 4;                // one workgroup per CU; the four cout quarters of a run on neighbouring workgroups
    if (groups < 1) groups = 1;
    a.k = rs64::runs_per_image(B * a.ns, a.nu, groups);
    const long long nruns = (long long)B * a.ns * a.k;
    if (nruns < groups) groups = (int)nruns;
    conv_rs64_kernel<0, 128><<<4 * groups, 256, rs64::lds_bytes128(a.nseg), st>>>(a);
    return 0;
}

}  // namespace xfh
