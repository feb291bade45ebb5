// conv_bx_kernel<24, 24> and conv_bxs2_kernel<24> (csrc/k_conv_bx.hip: the 24-channel layers block2.0 / block2.1 and the stride-2 24 -> 64 layer block3.0; sliced out of
// the product source by tests/test_kernels_emulated.py into conv_bx24_slice.hpp, with the weight split helpers of api.hip in weight_split_slice.hpp) on the host.
// stdin: {B, H, W, stride (1 | 2; 3 = stride 1 on conv_bx_kernel, the round-4 form that the trace build still uses), relu, grid} int32, then in (B*24*H*W), w (cout*24*9; cout = 24 | 64), bias (cout) as fp32 (BatchNorm folded);
// stdout: out (B*cout*Ho*Wo), status (int32).
#include "emu.hpp"
#include <cstdio>
namespace xfh {
#include "weight_split_slice.hpp"
#include "bx_split_slice.hpp"
#include "conv_bx24_slice.hpp"
// the operand-order weight images, restated from the layout comments of BxArgs / BxS2Args (the product packs them in xfh_create):
//   [cout block (stride 2 only)][step][fragment 3][lane = half * 32 + cout][8]: K group kg = 2 step + half = (tap kg / 3, channel group kg % 3), zero beyond group 26 / cout
static void pack24(const float* w, int cout, uint16_t* dst) {
    const int nstep = 14, ncb = cout > 32 ? 2 : 1;
    for (int cb = 0; cb < ncb; ++cb)
        for (int st = 0; st < nstep; ++st)
            for (int lane = 0; lane < 64; ++lane) {
                const int o = cb * 32 + (lane & 31), kg = 2 * st + (lane >> 5);
                for (int i = 0; i < 8; ++i) {
                    float v = 0.f;
                    if (o < cout && kg < 27) v = w[((size_t)o * 24 + (kg % 3) * 8 + i) * 9 + kg / 3];
                    uint16_t q[3];
                    split_weight(v, q);
                    for (int sp = 0; sp < 3; ++sp) dst[((((size_t)cb * nstep + st) * 3 + sp) * 64 + lane) * 8 + i] = q[sp];
                }
            }
}
}  // namespace xfh

static std::vector<float> rd(size_t n) {
    std::vector<float> v(n);
    if (fread(v.data(), 4, n, stdin) != n) { fprintf(stderr, "short input\n"); exit(2); }
    return v;
}

int main() {
    int hdr[6];
    if (fread(hdr, 4, 6, stdin) != 6) return 2;
    const int B = hdr[0], H = hdr[1], W = hdr[2], stride = hdr[3] == 3 ? 1 : hdr[3], relu = hdr[4], grid = hdr[5];
    const bool old_form = hdr[3] == 3;
    const int cout = stride == 2 ? 64 : 24, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    auto in = rd((size_t)B * 24 * H * W), w = rd((size_t)cout * 24 * 9);
    std::vector<float> bias = rd(cout);
    bias.resize(64, 0.f);
    std::vector<uint16_t> wq((size_t)2 * 14 * 3 * 64 * 8);
    xfh::pack24(w.data(), cout, wq.data());
    std::vector<float> out((size_t)B * cout * Ho * Wo, NAN);
    int status = 0;
    const int tiles_x = (W + 31) / 32, tiles = tiles_x * ((H + 7) / 8);
    const int total = tiles * B, g = total < grid ? total : grid;
    if (stride == 1 && !old_form) {      // what ships: conv_bxd_kernel (16 x 32 tiles, eight waves, two tile buffers)
        xfh::BxArgs a{};
        a.in = in.data(); a.wfrag = reinterpret_cast<const uint4*>(wq.data()); a.bias = bias.data(); a.out = out.data(); a.relu = relu; a.H = H; a.W = W; a.B = B;
        a.tiles_x = tiles_x; a.tiles = tiles_x * ((H + 15) / 16); a.lag = 0; a.status = &status;
        const int tot = a.tiles * B, gd = tot < grid ? tot : grid;
        emu::launch(gd, 512, xfh::BxdCfg<24, 24>::LDS_BYTES, [&] { xfh::conv_bxd_kernel<24, 24>(a); });
    } else if (stride == 1) {
        xfh::BxArgs a{};
        a.in = in.data(); a.wfrag = reinterpret_cast<const uint4*>(wq.data()); a.bias = bias.data(); a.out = out.data(); a.relu = relu; a.H = H; a.W = W; a.B = B;
        a.tiles_x = tiles_x; a.tiles = tiles; a.lag = 0; a.status = &status;
        emu::launch(g, 256, xfh::BxCfg<24, 24>::LDS_BYTES, [&] { xfh::conv_bx_kernel<24, 24>(a); });
    } else {
        xfh::BxS2Args a{};
        a.in = in.data(); a.wfrag = reinterpret_cast<const uint4*>(wq.data()); a.bias = bias.data(); a.out = out.data(); a.relu = relu; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.B = B;
        a.tiles_x = tiles_x; a.tiles = tiles; a.status = &status;
        emu::launch(g, 256, xfh::BxS2Cfg<24>::LDS_BYTES, [&] { xfh::conv_bxs2_kernel<24>(a); });
    }
    fwrite(out.data(), 4, out.size(), stdout);
    fwrite(&status, 4, 1, stdout);
    return 0;
}
