// block1_fused_kernel<5> (csrc/k_conv_direct.hip, sliced out of the product source by tests/test_kernels_emulated.py into block1_slice.hpp) on the host.
// stdin = {B, H, W, mode} int32, then gray (B*H*W), coef (2B), w1 (36), b1 (4), w2 (288), b2 (8), w3 (576), b3 (8), w4 (1728), b4 (32), skw (32), skb (32) as fp32 in the
// kernel's layouts (w_kc: [(ci * 9 + tap) * cout + co]); stdout = x1 (B*24*H/4*W/4 fp32).
#include "emu.hpp"
namespace xfh {
#include "block1_slice.hpp"
}
#include <cstdio>

static std::vector<float> rd(size_t n) {
    std::vector<float> v(n);
    if (fread(v.data(), 4, n, stdin) != n) { fprintf(stderr, "short input\n"); exit(2); }
    return v;
}

int main() {
    int hdr[4];
    if (fread(hdr, 4, 4, stdin) != 4) return 2;
    const int B = hdr[0], H = hdr[1], W = hdr[2], mode = hdr[3];
    auto gray = rd((size_t)B * H * W), coef = rd(2 * B), w1 = rd(36), b1 = rd(4), w2 = rd(288), b2 = rd(8), w3 = rd(576), b3 = rd(8), w4 = rd(1728), b4 = rd(32), skw = rd(32), skb = rd(32);
    const int H4 = H / 4, W4 = W / 4, tx = (W4 + xfh::b1::OW - 1) / xfh::b1::OW, ty = (H4 + xfh::b1::OH - 1) / xfh::b1::OH;
    std::vector<float> x1((size_t)B * 24 * H4 * W4, NAN);
    auto run = [&](auto M) {
        constexpr int MODE = decltype(M)::value;
        emu::launch(tx * ty * B, 512, (size_t)(MODE == 5 ? xfh::b1::F_LDS_FLOATS : xfh::b1::LDS_FLOATS) * 4, [&] {
            xfh::block1_fused_kernel<MODE>(gray.data(), coef.data(), x1.data(), B, H, W, tx, ty, w1.data(), b1.data(), w2.data(), b2.data(), w3.data(), b3.data(), w4.data(), b4.data(),
                                           skw.data(), skb.data());
        });
    };
    if (mode == 5) run(std::integral_constant<int, 5>{});
    else if (mode == 4) run(std::integral_constant<int, 4>{});
    else if (mode == 1) run(std::integral_constant<int, 1>{});
    else return 3;
    fwrite(x1.data(), 4, x1.size(), stdout);
    return 0;
}
