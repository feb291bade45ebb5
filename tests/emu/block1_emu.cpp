// block1_fused_body<MODE> (csrc/block1_body.hpp) on the host: stdin = {B, H, W, mode} as int32, then gray (B*H*W), coef (2B), w1 (36), b1 (4), w2 (288), b2 (8),
// w3 (576), b3 (8), w4 (1728), b4 (32), skw (32), skb (32) as fp32 in the kernel's layouts (w_kc: [(ci * 9 + tap) * cout + co]); stdout = x1 (B*24*H/4*W/4 fp32), status (int32).
#include "emu.hpp"
#include "block1_body.hpp"
#include "weight_split.hpp"
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
    std::vector<uint16_t> w4fx(xfh::b1fx::W4_BYTES / 2), w3fx(xfh::b1fx::W3_IMAGE_BYTES / 2);
    auto split = [](float v, uint16_t (&q)[3]) { xfh::split_weight(v, q); };
    xfh::b1fx::pack_w4(w4.data(), w4fx.data(), split);
    xfh::b1fx::pack_w3(w3.data(), w3fx.data(), split);
    const int H4 = H / 4, W4 = W / 4, tx = (W4 + xfh::b1::OW - 1) / xfh::b1::OW, ty = (H4 + xfh::b1::OH - 1) / xfh::b1::OH;
    std::vector<float> x1((size_t)B * 24 * H4 * W4, NAN);
    int status = 0;
    const size_t lds = (mode == 7 ? xfh::b1::M_LDS_FLOATS : xfh::b1::F_LDS_FLOATS) * 4 + (mode == 7 ? xfh::b1fx::W3_BYTES : 0);
    auto run = [&](auto M) {
        emu::launch(tx * ty * B, 512, lds, [&] {
            xfh::block1_fused_body<decltype(M)::value>(gray.data(), coef.data(), x1.data(), B, H, W, tx, ty, w1.data(), b1.data(), w2.data(), b2.data(), w3.data(), b3.data(),
                                                        w4.data(), b4.data(), skw.data(), skb.data(), w4fx.data(), w3fx.data(), &status, 0);
        });
    };
    if (mode == 5) run(std::integral_constant<int, 5>{});
    else if (mode == 7) run(std::integral_constant<int, 7>{});
    else return 3;
    fwrite(x1.data(), 4, x1.size(), stdout);
    fwrite(&status, 4, 1, stdout);
    return 0;
}
