// conv_bx64_body<64, FUSE, FX> (csrc/conv_bx64_body.hpp) on the host.  stdin: {B, H, W, fuse, fx, relu, relu2, grid} int32, then in (B*64*H*W), w (64*64*9), bias (64),
// [fuse: w2 (64*64), bias2 (64)] as fp32 (BatchNorm folded); stdout: out (B*64*H*W; fuse 2: channels-last), status (int32).
#include "emu.hpp"
#include "weight_split.hpp"
#include "conv_bx64_body.hpp"
#include <cstdio>

static std::vector<float> rd(size_t n) {
    std::vector<float> v(n);
    if (fread(v.data(), 4, n, stdin) != n) { fprintf(stderr, "short input\n"); exit(2); }
    return v;
}

int main() {
    int hdr[8];
    if (fread(hdr, 4, 8, stdin) != 8) return 2;
    const int B = hdr[0], H = hdr[1], W = hdr[2], fuse = hdr[3], fx = hdr[4], relu = hdr[5], relu2 = hdr[6], grid = hdr[7];
    if (fuse == 3) {      // the split-format link: conv A (plain 3x3, writes fp16 pairs) -> conv B (3x3 + 1x1, channels-last, reads them).  stdin continues: in, wA, bA, wB, bB, w2, b2
        auto in = rd((size_t)B * 64 * H * W), wA = rd(64 * 64 * 9), bA = rd(64), wB = rd(64 * 64 * 9), bB = rd(64), w2 = rd(64 * 64), b2 = rd(64);
        std::vector<uint16_t> qA((size_t)4 * 9 * 2 * 3 * 64 * 8 + 8192), qB(qA.size()), q2((size_t)4 * 2 * 3 * 64 * 8);
        xfh::pack_bx64(wA.data(), 64, 64, 1, qA.data()); xfh::pack_bx64(wB.data(), 64, 64, 1, qB.data()); xfh::pack_bx1x1(w2.data(), 1, q2.data());
        std::vector<float> mid((size_t)B * 64 * H * W, NAN), out((size_t)B * 64 * H * W, NAN), zeros(256, 0.f);
        int status = 0;
        xfh::Bx64Args a{};
        a.status = &status; a.zeros = zeros.data();
        a.in = in.data(); a.wq = qA.data(); a.bias = bA.data(); a.out = mid.data(); a.relu = relu; a.H = H; a.W = W; a.B = B;
        a.ncols = (W + 15) / 16; a.nhr = (H + 7) / 8; a.upi = a.ncols * a.nhr;
        const long long units = (long long)B * a.upi;
        const int g = units < grid ? (int)units : grid;
        emu::launch(g, 256, xfh::bx64::LDS_BYTES, [&] { xfh::conv_bx64_body<64, 0, 1, 2>(a); });
        xfh::Bx64Args b = a;
        b.in = mid.data(); b.wq = qB.data(); b.bias = bB.data(); b.out = out.data(); b.wq2 = reinterpret_cast<const uint4*>(q2.data()); b.bias2 = b2.data(); b.relu2 = relu2;
        emu::launch(g, 256, xfh::bx64::SP_LDS_BYTES, [&] { xfh::conv_bx64_body<64, 2, 1, 1>(b); });
        fwrite(out.data(), 4, out.size(), stdout);
        fwrite(&status, 4, 1, stdout);
        return 0;
    }
    auto in = rd((size_t)B * 64 * H * W), w = rd(64 * 64 * 9), bias = rd(64);
    std::vector<float> w2, bias2;
    if (fuse) { w2 = rd(64 * 64); bias2 = rd(64); }
    std::vector<uint16_t> wq((size_t)4 * 9 * 2 * 3 * 64 * 8 + 8192), wq2((size_t)4 * 2 * 3 * 64 * 8);
    xfh::pack_bx64(w.data(), 64, 64, fx ? 1 : 0, wq.data(), fx == 2 ? 2 : 3);
    if (fuse) xfh::pack_bx1x1(w2.data(), fx ? 1 : 0, wq2.data());
    std::vector<float> out((size_t)B * 64 * H * W, NAN);
    int status = 0;
    xfh::Bx64Args a{};
    a.status = &status;
    a.in = in.data(); a.wq = wq.data(); a.bias = bias.data(); a.out = out.data(); a.relu = relu; a.H = H; a.W = W; a.B = B;
    a.wq2 = fuse ? reinterpret_cast<const uint4*>(wq2.data()) : nullptr; a.bias2 = fuse ? bias2.data() : nullptr; a.relu2 = relu2;
    a.ncols = (W + 15) / 16; a.nhr = (H + 7) / 8; a.upi = a.ncols * a.nhr;
    const long long units = (long long)B * a.upi;
    const int g = units < grid ? (int)units : grid;
    auto run = [&](auto F, auto X) {
        emu::launch(g, 256, xfh::bx64::LDS_BYTES, [&] { xfh::conv_bx64_body<64, decltype(F)::value, decltype(X)::value>(a); });
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    auto runf = [&](auto F) { if (fx == 2) run(F, I2{}); else if (fx == 1) run(F, I1{}); else run(F, I0{}); };
    if (fuse == 0) runf(I0{}); else if (fuse == 1) runf(I1{}); else runf(I2{});
    fwrite(out.data(), 4, out.size(), stdout);
    fwrite(&status, 4, 1, stdout);
    return 0;
}
