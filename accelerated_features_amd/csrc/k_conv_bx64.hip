// 3x3 stride-1 convolution, 64 (or 128) -> 64 channels, on the bf16 matrix cores with three-way split operands (see k_conv_bx.hip for
// the arithmetic: six v_mfma_f32_32x32x16_bf16 per K=16 carry an fp32 product sum).     block3.1, block4.1/.2, block_fusion.0/.1
//
// The 24-channel kernel keeps its weights in registers; here the split weights are 216 KiB, so both operands come from LDS:
//   * a wave owns 2 pixel blocks x 2 cout blocks (four 32x32 accumulators): per K step 6 + 6 ds_read_b128 feed 24 MFMAs;
//   * pixel block = 2 rows x 16 columns (lane l31 -> row l31 >> 4, column l31 & 15); a workgroup (4 waves) owns a 16x16 tile, or an
//     8x16 "half tile" (one pixel block per wave).  The work list is cut in units of half tiles, so that every workgroup of the
//     persistent grid gets the same number of units (VGA batch 64 at 1/8 scale: 2560 units = 5 per workgroup = two tiles and a half
//     tile; whole tiles only would be 2.5 per workgroup: three rounds for half of the chip);
//   * the input channels go through LDS in chunks of 16 = one K step per tap: [18 halo rows, 2048 B apart][18 pixels, 112 B apart]
//     [split h, m, l][16 channels] bf16 -- 112 B = 7 x 16 B keeps the 16 lanes of every ds_read_b128 group on distinct banks, the
//     256-B-multiple row pitch does the same for the second row of a pixel block.  Raw fp32 values are prefetched into registers one
//     chunk ahead (also across tiles), split on the way into LDS;
//   * the weights stream through a two-slot LDS ring by LDS-DMA, one slot = one tap row of one chunk (3 K steps, 18 KiB, in operand
//     order [tap][cout block][split][lane]); the DMA of row r+1 is issued behind the barrier that opens row r;
//   * two workgroups per CU (74 KiB of LDS each) overlap each other's barriers, staging and stores.
#include "kernels.hpp"
#include "bx_split.hpp"
#include <cstdlib>
#include <type_traits>
#include "conv_bx64_body.hpp"

namespace xfh {

// body in conv_bx64_body.hpp (also compiled for the host by tests/emu/)
template <int CIN, int FUSE, int FXM, int SP = 0>      // FXM: 0 bf16 three-way split, 1 fp16 pair, 2 fp16 pair with two weight fragments in the stream; SP: 1 input / 2 output in the split format (conv_bx64_body.hpp)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv_bx64_kernel(Bx64Args a) {
    kernel_entry_hooks(a.cold);      // debug: code-position shift / cold instruction cache (common.hpp)
    conv_bx64_body<CIN, FUSE, FXM, SP>(a);
}

template <int CIN, int FUSE, int FXM, int SP = 0>
static int run_bx64(const ConvW& c, const ConvW* c2, const float* in, int B, int H, int W, float* out, hipStream_t st, long long* trace, int* status, const float* zeros = nullptr) {
    if ((size_t)CIN * H * W * sizeof(float) >= 0x7fffffffu || (size_t)64 * H * W * sizeof(float) >= 0x7fffffffu) return -1;      // buffer-resource range
    Bx64Args a;
    a.cold = g_debug_cold;
    a.status = status;
    a.zeros = zeros;
    a.in = in; a.wq = FXM == 2 ? c.w_fq : FXM ? c.w_fx : c.w_bx; a.bias = c.bias; a.out = out; a.relu = c.relu; a.H = H; a.W = W; a.B = B; a.trace = trace;
    a.wq2 = c2 ? reinterpret_cast<const uint4*>(FXM ? c2->w_fx : c2->w_bx) : nullptr; a.bias2 = c2 ? c2->bias : nullptr; a.relu2 = c2 ? c2->relu : 0;
    a.ncols = ceil_div(W, 16); a.nhr = ceil_div(H, 8); a.upi = a.ncols * a.nhr;
    static AttrMask attr_done = 0;
    constexpr int lds_bytes = (SP & 1) ? bx64::SP_LDS_BYTES : bx64::LDS_BYTES;
    set_max_dynamic_lds(reinterpret_cast<const void*>(conv_bx64_kernel<CIN, FUSE, FXM, SP>), lds_bytes, attr_done);
    const long long units = (long long)B * a.upi;
    int grid = 2 * num_cus();                  // two resident workgroups per CU; a multiple of 8 keeps a workgroup on its XCD
    if (units < grid) grid = (int)units;       // (small inputs: one unit per workgroup; the XCD mapping then needs grid % 8 == 0 or is skipped)
    conv_bx64_kernel<CIN, FUSE, FXM, SP><<<grid, 256, lds_bytes, st>>>(a);
    return 0;
}

int launch_conv_bx64(const ConvW& c, const float* in, int B, int H, int W, float* out, hipStream_t st, long long* trace, const ConvW* c2, bool nhwc, int fx, int* status, int sp, const float* zeros) {
    if (c.ks != 3 || c.stride != 1 || !c.w_bx || c.cout != 64 || c.cin != 64) return -1;
    if (c2 && (c2->ks != 1 || c2->cin != 64 || c2->cout != 64 || !c2->w_bx)) return -1;
    if (sp) {      // the split-format link (conv_bx64_body.hpp): 2 = the plain 3x3 writes it, 1 = the fused channels-last form reads it; fp16 pair, three weight fragments
        if (!(fx && c.w_fx) || !zeros) return -1;
        if (sp == 2 && !c2 && !nhwc) return run_bx64<64, 0, 1, 2>(c, nullptr, in, B, H, W, out, st, trace, status, zeros);
        if (sp == 1 && c2 && c2->w_fx && nhwc) return run_bx64<64, 2, 1, 1>(c, c2, in, B, H, W, out, st, trace, status, zeros);
        return -1;
    }
    if (fx && c.w_fx && (!c2 || c2->w_fx)) {      // the fp16-pair arithmetic: three MFMAs per product instead of six; fx = 2: two weight fragments in the stream
        if (fx == 2 && c.w_fq) {
            if (!c2) return nhwc ? -1 : run_bx64<64, 0, 2>(c, nullptr, in, B, H, W, out, st, trace, status);
            return nhwc ? run_bx64<64, 2, 2>(c, c2, in, B, H, W, out, st, trace, status) : run_bx64<64, 1, 2>(c, c2, in, B, H, W, out, st, trace, status);
        }
        if (!c2) return nhwc ? -1 : run_bx64<64, 0, 1>(c, nullptr, in, B, H, W, out, st, trace, status);
        return nhwc ? run_bx64<64, 2, 1>(c, c2, in, B, H, W, out, st, trace, status) : run_bx64<64, 1, 1>(c, c2, in, B, H, W, out, st, trace, status);
    }
    if (!c2) return nhwc ? -1 : run_bx64<64, 0, 0>(c, nullptr, in, B, H, W, out, st, trace, status);
    return nhwc ? run_bx64<64, 2, 0>(c, c2, in, B, H, W, out, st, trace, status) : run_bx64<64, 1, 0>(c, c2, in, B, H, W, out, st, trace, status);
}

}  // namespace xfh
