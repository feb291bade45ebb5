// conv_rs64_body (csrc/conv_rs64_body.hpp) on the host.  stdin: {B, H, W, relu, grid, k} int32, then in (B*64*H*W), w (64*64*9), bias (64) as fp32 (BatchNorm folded);
// stdout: out (B*64*H*W), status (int32).  k = runs per image (0: the launcher's choice for `grid` workgroups)
#include "emu.hpp"
#include "weight_split.hpp"
#include "conv_rs64_body.hpp"
#include <cstdio>

static std::vector<float> rd(size_t n) {
    std::vector<float> v(n);
    if (fread(v.data(), 4, n, stdin) != n) { fprintf(stderr, "short input\n"); exit(2); }
    return v;
}

int main() {
    int hdr[6];
    if (fread(hdr, 4, 6, stdin) != 6) return 2;
    const int B = hdr[0], H = hdr[1], W = hdr[2], relu = hdr[3], grid = hdr[4];
    auto in = rd((size_t)B * 64 * H * W), w = rd(64 * 64 * 9), bias = rd(64);
    std::vector<uint16_t> wq(xfh::rs64::WQ_HALFS);
    xfh::pack_rs64(w.data(), wq.data());
    std::vector<float> out((size_t)B * 64 * H * W, NAN);
    int status = 0;
    xfh::Rs64Args a{};
    a.in = in.data(); a.wq = wq.data(); a.bias = bias.data(); a.out = out.data(); a.relu = relu; a.H = H; a.W = W; a.B = B; a.status = &status;
    a.P = W + 2; a.inv_p = 1.f / (float)a.P; a.nu = (H * a.P + 63) / 64; a.nseg = xfh::rs64::nseg_for(a.P);
    if (a.nseg > xfh::rs64::MAX_NSEG) { fprintf(stderr, "map too wide\n"); return 3; }
    a.k = hdr[5] > 0 ? hdr[5] : xfh::rs64::runs_per_image(B, a.nu, grid);
    const int nruns = B * a.k, g = nruns < grid ? nruns : grid;
    emu::launch(g, 256, xfh::rs64::lds_bytes(a.nseg), [&] { xfh::conv_rs64_body(a); });
    fwrite(out.data(), 4, out.size(), stdout);
    fwrite(&status, 4, 1, stdout);
    return 0;
}
