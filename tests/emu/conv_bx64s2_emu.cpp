// conv_bx64s2w_body<NCO, W4> (csrc/conv_bx64s2_body.hpp: twelve waves, eight multiplying + four staging) on the host.  stdin: {B, H, W, cout (64 | 128), relu, grid} int32, then in (B*64*H*W), w (cout*64*9), bias (cout)
// as fp32 (BatchNorm folded); stdout: out (B*cout*Ho*Wo), status (int32).
#include "emu.hpp"
#define XFH_S2_KEEP5(a, b, c, d, e) ((void)0)
#include "weight_split.hpp"
#include "conv_bx64s2_body.hpp"
#include <cstdio>

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
    xfh::pack_bx64(w.data(), 64, cout, wq.data());
    std::vector<float> out((size_t)B * cout * Ho * Wo, NAN);
    int status = 0;
    xfh::Bx64S2xArgs a{};
    a.status = &status;
    a.in = in.data(); a.wq = wq.data(); a.bias = bias.data(); a.out = out.data(); a.relu = relu; a.H = H; a.W = W; a.Ho = Ho; a.Wo = Wo; a.B = B;
    a.nrows = (Ho + 7) / 8; a.upi = ((Wo + 15) / 16) * a.nrows;
    const long long units = (long long)nco * B * a.upi;
    const int g = units < grid ? (int)units : grid;
    const bool w4 = (W & 3) == 0;
    auto run = [&](auto N, auto W4) { emu::launch(g, 768, xfh::bx64s2x::LDS_W_BYTES, [&] { xfh::conv_bx64s2w_body<decltype(N)::value, decltype(W4)::value>(a); }); };
    using T = std::true_type; using F = std::false_type; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    auto r1 = [&](auto N) { if (w4) run(N, T{}); else run(N, F{}); };
    if (nco == 1) r1(I1{}); else r1(I2{});
    fwrite(out.data(), 4, out.size(), stdout);
    fwrite(&status, 4, 1, stdout);
    return 0;
}
