// conv_bx_kernel<24, 24> and conv_bxs2_kernel<24> (csrc/k_conv_bx.hip: the 24-channel layers block2.0 / block2.1 and the stride-2 24 -> 64 layer block3.0; sliced out of
// the product source by tests/test_kernels_emulated.py into conv_bx24_slice.hpp, with the weight split helpers of api.hip in weight_split_slice.hpp) on the host.
// stdin: {B, H, W, stride (1 | 2; 3 = stride 1 on conv_bx_kernel, the round-4 form that the trace build still uses; 5 / 7 / 8: the channels-last forms of the backbone's links, below), relu, grid} int32, then in (B*24*H*W), w (cout*24*9; cout = 24 | 64), bias (cout) as fp32 (BatchNorm folded);
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
    // layout codes on top of the stride: 5 = stride 1, planes in -> channels-last out; 7 = stride 1, channels-last in and out; 8 = stride 2, channels-last in (the backbone's links
    // block2.0 -> block2.1 -> block3.0).  The transposition to and from planes is done here: the test feeds and reads planes whatever the code.
    const int code = hdr[3];
    const bool in_cl = code == 7 || code == 8, out_cl = code == 5 || code == 7;
    const int B = hdr[0], H = hdr[1], W = hdr[2], stride = (code == 3 || code == 5 || code == 7) ? 1 : (code == 8 ? 2 : code), relu = hdr[4], grid = hdr[5];
    const bool old_form = code == 3;
    const int cout = stride == 2 ? 64 : 24, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    auto in = rd((size_t)B * 24 * H * W), w = rd((size_t)cout * 24 * 9);
    std::vector<float> bias = rd(cout);
    bias.resize(64, 0.f);
    std::vector<uint16_t> wq((size_t)2 * 14 * 3 * 64 * 8);
    xfh::pack24(w.data(), cout, wq.data());
    std::vector<float> out((size_t)B * cout * Ho * Wo, NAN);
    if (in_cl) {      // (B, 24, H, W) -> (B, H, W, 24)
        std::vector<float> t(in.size());
        for (int b = 0; b < B; ++b) for (int c = 0; c < 24; ++c) for (int p = 0; p < H * W; ++p) t[((size_t)b * H * W + p) * 24 + c] = in[((size_t)b * 24 + c) * H * W + p];
        in.swap(t);
    }
    int status = 0;
    const int tiles_x = (W + 31) / 32, tiles = tiles_x * ((H + 7) / 8);
    const int total = tiles * B, g = total < grid ? total : grid;
    if (stride == 1 && !old_form) {      // what ships: conv_bxd_kernel (16 x 32 tiles, eight waves, two tile buffers)
        xfh::BxArgs a{};
        a.in = in.data(); a.wfrag = reinterpret_cast<const uint4*>(wq.data()); a.bias = bias.data(); a.out = out.data(); a.relu = relu; a.H = H; a.W = W; a.B = B;
        a.tiles_x = tiles_x; a.tiles = tiles_x * ((H + 15) / 16); a.lag = 0; a.status = &status;
        const int tot = a.tiles * B, gd = tot < grid ? tot : grid;
        if (in_cl) emu::launch(gd, 512, xfh::BxdCfg<24, 24>::LDS_BYTES, [&] { xfh::conv_bxd_kernel<24, 24, true, true>(a); });
        else if (out_cl) emu::launch(gd, 512, xfh::BxdCfg<24, 24>::LDS_BYTES, [&] { xfh::conv_bxd_kernel<24, 24, false, true>(a); });
        else emu::launch(gd, 512, xfh::BxdCfg<24, 24>::LDS_BYTES, [&] { xfh::conv_bxd_kernel<24, 24>(a); });
    } else if (stride == 1) {
        xfh::BxArgs a{};
        a.in = in.data(); a.wfrag = reinterpret_cast<const uint4*>(wq.data()); a.bias = bias.data(); a.out = out.data(); a.relu = relu; a.H = H; a.W = W; a.B = B;
        a.tiles_x = tiles_x; a.tiles = tiles; a.lag = 0; a.status = &status;
        emu::launch(g, 256, xfh::BxCfg<24, 24>::LDS_BYTES, [&] { xfh::conv_bx_kernel<24, 24>(a); });
    } else {
        xfh::BxS2Args a{};
        a.in = in.data(); a.wfrag = reinterpret_cast<const uint4*>(wq.data()); a.bias = bias.data(); a.out = out.data(); a.relu = relu; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.B = B;
        a.tiles_x = tiles_x; a.tiles = tiles; a.status = &status;
        if (in_cl) emu::launch(g, 256, xfh::BxS2Cfg<24>::LDS_BYTES, [&] { xfh::conv_bxs2_kernel<24, true>(a); });
        else emu::launch(g, 256, xfh::BxS2Cfg<24>::LDS_BYTES, [&] { xfh::conv_bxs2_kernel<24>(a); });
    }
    if (out_cl) {      // (B, Ho, Wo, 24) -> (B, 24, Ho, Wo)
        std::vector<float> t(out.size());
        for (int b = 0; b < B; ++b) for (int c = 0; c < cout; ++c) for (int p = 0; p < Ho * Wo; ++p) t[((size_t)b * cout + c) * Ho * Wo + p] = out[((size_t)b * Ho * Wo + p) * cout + c];
        out.swap(t);
    }
    fwrite(out.data(), 4, out.size(), stdout);
    fwrite(&status, 4, 1, stdout);
    return 0;
}
