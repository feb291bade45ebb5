// head_f32r_kernel<KP> (csrc/k_heads.hip: the default heads, sliced out of the product source by tests/test_kernels_emulated.py into heads_slice.hpp) on the host.
// stdin: {kp, B, H, W} int32 (REL: B = cells, H = W = 0), then
//   KP : gray (B*H*W), coef (2B), w0..w2 (64x64 each), w3 (65x64), b0..b2 (64 each), b3 (65)         -> stdout heat (B*H*W), logits (cells*65)
//   REL: feats (cells*64), w0, w1 (64x64), w2 (64), b0, b1 (64), b2 (1)                               -> stdout reliability (cells), inv (cells)
#include "emu.hpp"
namespace xfh {
#include "heads_slice.hpp"
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
    const int kp = hdr[0], B = hdr[1], H = hdr[2], W = hdr[3];
    xfh::HeadArgs f{};
    // the kernels' weight layout: [k][n_pad] (BN folded), n_pad = 64 | 96
    auto kcp = [](const std::vector<float>& w, int cout, int npad) { std::vector<float> o((size_t)64 * npad, 0.f); for (int n = 0; n < cout; ++n) for (int k = 0; k < 64; ++k) o[(size_t)k * npad + n] = w[(size_t)n * 64 + k]; return o; };
    if (kp) {
        auto gray = rd((size_t)B * H * W), coef = rd(2 * B);
        std::vector<std::vector<float>> ws = {rd(4096), rd(4096), rd(4096), rd(65 * 64)}, bs = {rd(64), rd(64), rd(64), rd(65)};
        std::vector<std::vector<float>> wk = {kcp(ws[0], 64, 64), kcp(ws[1], 64, 64), kcp(ws[2], 64, 64), kcp(ws[3], 65, 96)};
        bs[3].resize(96, 0.f);
        for (int i = 0; i < 4; ++i) { f.w[i] = wk[i].data(); f.bias[i] = bs[i].data(); }
        f.src = gray.data(); f.coef = coef.data(); f.H = H; f.W = W; f.hc = H / 8; f.wc = W / 8; f.ncell = B * f.hc * f.wc; f.ntiles = (f.ncell + 255) / 256;
        std::vector<float> heat((size_t)B * H * W, NAN), logits((size_t)f.ncell * 65, NAN);
        f.out = heat.data(); f.logits = logits.data();
        emu::launch(std::min(f.ntiles, 2), 512, (size_t)(3 * 64 * 64 + 64 * 96) * 4, [&] { xfh::head_f32r_kernel<true>(f); });
        fwrite(heat.data(), 4, heat.size(), stdout);
        fwrite(logits.data(), 4, logits.size(), stdout);
    } else {
        auto feats = rd((size_t)B * 64);
        std::vector<std::vector<float>> ws = {rd(4096), rd(4096)};
        auto w2 = rd(64);
        std::vector<std::vector<float>> bs = {rd(64), rd(64)};
        auto b2 = rd(1);
        std::vector<std::vector<float>> wk = {kcp(ws[0], 64, 64), kcp(ws[1], 64, 64)};
        f.w[0] = wk[0].data(); f.w[1] = wk[1].data(); f.w[2] = w2.data(); f.bias[0] = bs[0].data(); f.bias[1] = bs[1].data(); f.bias[2] = b2.data();
        f.src = feats.data(); f.hc = 1; f.wc = 1; f.H = 8; f.W = 8; f.ncell = B; f.ntiles = (B + 255) / 256;
        std::vector<float> rel(B, NAN), inv(B, NAN);
        f.out = rel.data(); f.inv = inv.data();
        emu::launch(std::min(f.ntiles, 2), 512, (size_t)(2 * 64 * 64 + 64) * 4, [&] { xfh::head_f32r_kernel<false>(f); });
        fwrite(rel.data(), 4, rel.size(), stdout);
        fwrite(inv.data(), 4, inv.size(), stdout);
    }
    return 0;
}
