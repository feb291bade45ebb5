// conv_bx64s2_kernel<NCO, W4> (csrc/k_conv_bx64s2.hip: the stride-2 64 -> 64 | 128 convolutions, block4.0 / block5.0; sliced out of the product source by
// tests/test_kernels_emulated.py into conv_bx64s2_slice.hpp, with the weight split helpers of api.hip in weight_split_slice.hpp) on the host.
// stdin: {B, H, W, cout (64 | 128), relu, grid} int32, then in (B*64*H*W), w (cout*64*9), bias (cout) as fp32 (BatchNorm folded); stdout: out (B*cout*Ho*Wo).
#include "emu.hpp"
#include <cstdio>
namespace xfh {
#include "weight_split_slice.hpp"
#include "bx_split_slice.hpp"
#include "conv_bx64s2_slice.hpp"
// the operand-order weight image, restated from the layout comment of Bx64S2Args (the product packs it in xfh_create):
//   [cout half][cin/16][tap 9][cout block 2][split 3][lane = half * 32 + cout][8], channel = 16 chunk + 8 half + i
static void pack3x3(const float* w, int cout, uint16_t* dst) {
    for (int hf = 0; hf < cout / 64; ++hf)
        for (int ch = 0; ch < 4; ++ch)
            for (int tap = 0; tap < 9; ++tap)
                for (int cb = 0; cb < 2; ++cb)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int i = 0; i < 8; ++i) {
                            const int o = hf * 64 + cb * 32 + (lane & 31), ci = ch * 16 + 8 * (lane >> 5) + i;
                            uint16_t q[3];
                            split_weight(w[((size_t)o * 64 + ci) * 9 + tap], 0, q);
                            for (int sp = 0; sp < 3; ++sp) dst[((((((size_t)hf * 4 + ch) * 9 + tap) * 2 + cb) * 3 + sp) * 64 + lane) * 8 + i] = q[sp];
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
    const int B = hdr[0], H = hdr[1], W = hdr[2], cout = hdr[3], relu = hdr[4], grid = hdr[5];
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, nco = cout / 64;
    auto in = rd((size_t)B * 64 * H * W), w = rd((size_t)cout * 64 * 9), bias = rd(cout);
    std::vector<uint16_t> wq((size_t)nco * 4 * 9 * 2 * 3 * 64 * 8 + 8192);
    xfh::pack3x3(w.data(), cout, wq.data());
    std::vector<float> out((size_t)B * cout * Ho * Wo, NAN);
    xfh::Bx64S2Args a{};
    a.in = in.data(); a.wq = wq.data(); a.bias = bias.data(); a.out = out.data(); a.relu = relu; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.B = B;
    a.nrows = (Ho + 7) / 8; a.upi = ((Wo + 15) / 16) * a.nrows;
    const long long units = (long long)nco * B * a.upi;
    const int g = units < grid ? (int)units : grid;
    const bool w4 = (W & 3) == 0;
    if (nco == 1) { if (w4) emu::launch(g, 512, xfh::bx64s2::LDS_BYTES, [&] { xfh::conv_bx64s2_kernel<1, true>(a); }); else emu::launch(g, 512, xfh::bx64s2::LDS_BYTES, [&] { xfh::conv_bx64s2_kernel<1, false>(a); }); }
    else { if (w4) emu::launch(g, 512, xfh::bx64s2::LDS_BYTES, [&] { xfh::conv_bx64s2_kernel<2, true>(a); }); else emu::launch(g, 512, xfh::bx64s2::LDS_BYTES, [&] { xfh::conv_bx64s2_kernel<2, false>(a); }); }
    fwrite(out.data(), 4, out.size(), stdout);
    return 0;
}
