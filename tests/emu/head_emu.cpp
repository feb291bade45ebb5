// head_bx_body<KP> (csrc/head_bx_body.hpp: the default heads, fp16-pair arithmetic; fx = 1) and head_f32r_body<KP> (csrc/head_f32r_body.hpp: the f32-MFMA fallback; fx = -1) on the host.
// stdin: {kp, fx, B, H, W} int32 (REL: B = cells, H = W = 0), then
//   KP : gray (B*H*W), coef (2B), w0..w2 (64x64 each), w3 (65x64), b0..b2 (64 each), b3 (65)         -> stdout heat (B*H*W), logits (cells*65), status
//   REL: feats (cells*64), w0, w1 (64x64), w2 (64), b0, b1 (64), b2 (1)                               -> stdout reliability (cells), inv (cells), status
#include "emu.hpp"
#include "weight_split.hpp"
#include "head_bx_body.hpp"
#include "head_f32r_body.hpp"
#include <cstdio>

static std::vector<float> rd(size_t n) {
    std::vector<float> v(n);
    if (fread(v.data(), 4, n, stdin) != n) { fprintf(stderr, "short input\n"); exit(2); }
    return v;
}

int main() {
    int hdr[5];
    if (fread(hdr, 4, 5, stdin) != 5) return 2;
    const int kp = hdr[0], fx = hdr[1], B = hdr[2], H = hdr[3], W = hdr[4];
    int status = 0;
    if (fx < 0) {      // the f32-MFMA heads with register input: weights [k][n_pad] (BN folded), n_pad = 64 | 96
        xfh::HeadArgs f{};
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
            emu::launch(std::min(f.ntiles, 2), 512, (size_t)(3 * 64 * 64 + 64 * 96) * 4, [&] { xfh::head_f32r_body<true>(f); });
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
            emu::launch(std::min(f.ntiles, 2), 512, (size_t)(2 * 64 * 64 + 64) * 4, [&] { xfh::head_f32r_body<false>(f); });
            fwrite(rel.data(), 4, rel.size(), stdout);
            fwrite(inv.data(), 4, inv.size(), stdout);
        }
        fwrite(&status, 4, 1, stdout);
        return 0;
    }
    if (fx != 1) return 3;
    xfh::HeadBxArgs a{};
    a.status = &status;
    std::vector<uint16_t> wq;
    std::vector<float> bias;
    auto pack = [&](const std::vector<std::vector<float>>& ws, const std::vector<std::vector<float>>& bs, const std::vector<int>& couts) {
        size_t words = 0;
        for (int c : couts) words += (size_t)4 * ((c + 31) / 32) * 3 * 64 * 8;
        wq.assign(words, 0);
        uint16_t* dst = wq.data();
        for (size_t p = 0; p < couts.size(); ++p) {
            dst += xfh::pack_head_layer(ws[p].data(), couts[p], p == 0, dst);
            const int nb = (int)bs[p].size(), pad = 32 * ((nb + 31) / 32);          // (the bias table keeps the layout 64, 64, 64, 96)
            for (int o = 0; o < pad; ++o) bias.push_back(o < nb ? bs[p][o] : 0.f);
        }
        a.wq = reinterpret_cast<const uint4*>(wq.data());
        a.bias = bias.data();
    };
    if (kp) {
        auto gray = rd((size_t)B * H * W), coef = rd(2 * B);
        std::vector<std::vector<float>> ws = {rd(4096), rd(4096), rd(4096), rd(65 * 64)}, bs = {rd(64), rd(64), rd(64), rd(65)};
        pack(ws, bs, {64, 64, 64, 64});          // (the dustbin logit is a dot product on the vector ALUs: w_dust / b_dust)
        a.w_dust = ws[3].data() + 64 * 64; a.b_dust = bs[3][64];
        a.src = gray.data(); a.coef = coef.data();
        a.H = H; a.W = W; a.hc = H / 8; a.wc = W / 8; a.ncell = B * a.hc * a.wc; a.ntiles = (a.ncell + 255) / 256;
        std::vector<float> heat((size_t)B * H * W, NAN), logits((size_t)a.ncell * 65, NAN);
        a.out = heat.data(); a.logits = logits.data();
        const size_t lds = (size_t)8 * 4 * 3 * 1024 + (288 + 64) * 4;
        const int grid = std::min(a.ntiles, 2);          // (persistent: each workgroup walks several tiles)
        emu::launch(grid, 512, lds, [&] { xfh::head_bx_body<true>(a); });
        fwrite(heat.data(), 4, heat.size(), stdout);
        fwrite(logits.data(), 4, logits.size(), stdout);
    } else {
        auto feats = rd((size_t)B * 64);
        std::vector<std::vector<float>> ws = {rd(4096), rd(4096)};
        auto w2 = rd(64);
        std::vector<std::vector<float>> bs = {rd(64), rd(64)};
        auto b2 = rd(1);
        pack(ws, bs, {64, 64});
        a.src = feats.data(); a.w_last = w2.data(); a.b_last = b2[0];
        a.hc = 1; a.wc = 1; a.H = 8; a.W = 8; a.ncell = B; a.ntiles = (B + 255) / 256;
        std::vector<float> rel(B, NAN), inv(B, NAN);
        a.out = rel.data(); a.inv = inv.data();
        const size_t lds = (size_t)2 * 2 * 4 * 3 * 1024 + (128 + 64) * 4;
        const int grid = std::min(a.ntiles, 2);
        emu::launch(grid, 512, lds, [&] { xfh::head_bx_body<false>(a); });
        fwrite(rel.data(), 4, rel.size(), stdout);
        fwrite(inv.data(), 4, inv.size(), stdout);
    }
    fwrite(&status, 4, 1, stdout);
    return 0;
}
