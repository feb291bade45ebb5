// conv_wino_kernel<...> (csrc/k_conv_wino.hip: Winograd F(2x2,3x3) on the f32 matrix cores -- block5.1, block5.2 + 5.3 on the default path, every 3x3/s1 layer under option
// wino; sliced out of the product source by tests/test_kernels_emulated.py into conv_wino_slice.hpp) on the host.
// stdin: {B, H, W, cin (64 | 128), fuse (0: the 3x3 alone | 1: + trailing 1x1 to 64, NCHW | 2: the same, channels-last), relu, relu2, tall (tile region 8x4 instead of 4x8)}
// int32, then in (B*cin*H*W), w (cin*cin*9), bias (cin) as fp32 (BatchNorm folded) [, w2 (64*cin), bias2 (64)]; stdout: out.
#include "emu.hpp"
#include <cstdio>
#include <type_traits>
#define __fmaf_rn(a, b, c) fmaf(a, b, c)
namespace xfh {
#include "conv_wino_slice.hpp"
// U = G g G^T in fp64, rounded once: [cin/4][pos = 4*xi + nu][half = ci & 1][cout_pad][p = (ci >> 1) & 1] -- restated from the layout comment of WinoArgs (the product
// packs it in xfh_create)
static void pack_wino(const float* w, int cin, int cout, float* dst) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int cpad = (cout + 31) / 32 * 32;
    for (int o = 0; o < cout; ++o)
        for (int i = 0; i < cin; ++i)
            for (int xi = 0; xi < 4; ++xi)
                for (int nu = 0; nu < 4; ++nu) {
                    double u = 0;
                    for (int r = 0; r < 3; ++r)
                        for (int s = 0; s < 3; ++s) u += G[xi][r] * (double)w[((size_t)o * cin + i) * 9 + r * 3 + s] * G[nu][s];
                    dst[((((size_t)(i / 4) * 16 + xi * 4 + nu) * 2 + (i & 1)) * cpad + o) * 2 + ((i >> 1) & 1)] = (float)u;
                }
}
template <int CIN, int CB, int TTH, int TTW, int COUT2, bool NHWC>
static void run(WinoArgs a) {
    using Cfg = WinoCfg<CB, 1, 2, 1, TTH, TTW, COUT2>;
    a.tiles_x = (((a.W + 1) / 2) + TTW - 1) / TTW;
    a.tiles = a.tiles_x * ((((a.H + 1) / 2) + TTH - 1) / TTH);
    emu::launch(a.tiles * a.B, Cfg::NTHR, (size_t)Cfg::LDS_FLOATS * sizeof(float), [&] { conv_wino_kernel<CIN, CIN, CB, 1, 2, 1, TTH, TTW, COUT2, NHWC>(a); });
}
}  // namespace xfh

static std::vector<float> rd(size_t n) {
    std::vector<float> v(n);
    if (fread(v.data(), 4, n, stdin) != n) { fprintf(stderr, "short input\n"); exit(2); }
    return v;
}

int main() {
    int hdr[8];
    if (fread(hdr, 4, 8, stdin) != 8) return 2;
    const int B = hdr[0], H = hdr[1], W = hdr[2], cin = hdr[3], fuse = hdr[4], tall = hdr[7];
    auto in = rd((size_t)B * cin * H * W), w = rd((size_t)cin * cin * 9), bias = rd(cin);
    std::vector<float> wu((size_t)cin * 16 * cin, 0.f), wk2, bias2;
    xfh::pack_wino(w.data(), cin, cin, wu.data());
    if (fuse) {      // the 1x1 as [cin][64] (= ConvW::w_kcp of a 1x1 layer: [ci][cout_pad])
        auto w2 = rd((size_t)64 * cin); bias2 = rd(64);
        wk2.assign((size_t)cin * 64, 0.f);
        for (int o = 0; o < 64; ++o) for (int i = 0; i < cin; ++i) wk2[(size_t)i * 64 + o] = w2[(size_t)o * cin + i];
    }
    std::vector<float> out((size_t)B * (fuse ? 64 : cin) * H * W, NAN);
    xfh::WinoArgs a{};
    a.in = in.data(); a.wu = wu.data(); a.bias = bias.data(); a.out = out.data(); a.relu = hdr[5]; a.relu2 = hdr[6]; a.H = H; a.W = W; a.B = B;
    a.wk2 = fuse ? wk2.data() : nullptr; a.bias2 = fuse ? bias2.data() : nullptr;
#define GO(CIN, CB, C2, NH) { if (tall) xfh::run<CIN, CB, 8, 4, C2, NH>(a); else xfh::run<CIN, CB, 4, 8, C2, NH>(a); }
    if (cin == 64 && fuse == 0) GO(64, 2, 0, false)
    else if (cin == 64 && fuse == 1) GO(64, 2, 64, false)
    else if (cin == 64 && fuse == 2) GO(64, 2, 64, true)
    else if (cin == 128 && fuse == 0) GO(128, 4, 0, false)
    else if (cin == 128 && fuse == 1) GO(128, 4, 64, false)
    else return 3;
    fwrite(out.data(), 4, out.size(), stdout);
    return 0;
}
