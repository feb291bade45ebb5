// conv_rs64_body (csrc/conv_rs64_body.hpp) on the host.  stdin: {B, H, W, relu, grid, k, fuse, relu2} int32, then in (B*64*H*W), w (64*64*9), bias (64) as fp32 (BatchNorm folded);
// [fuse: w2 (64*64), bias2 (64)]; stdout: out (B*64*H*W; fuse 2: channels-last), status (int32).  k = runs per image (0: the launcher's choice for `grid` workgroups)
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
    int hdr[8];
    if (fread(hdr, 4, 8, stdin) != 8) return 2;
    const int B = hdr[0], H = hdr[1], W = hdr[2], relu = hdr[3], grid = hdr[4], fuse = hdr[6];
    if (fuse == 128) {      // the 128 -> 128 form: in (B*128*H*W), w (128*128*9), bias (128); grid = workgroup GROUPS (x 4 cout quarters)
        auto in = rd((size_t)B * 128 * H * W), w = rd(128 * 128 * 9), bias = rd(128);
        std::vector<uint16_t> wq(4 * xfh::kRs64Halfs);
        xfh::pack_rs128(w.data(), wq.data());
        std::vector<float> out((size_t)B * 128 * H * W, NAN);
        int status = 0;
        xfh::Rs64Args a{};
        a.in = in.data(); a.wq = wq.data(); a.bias = bias.data(); a.out = out.data(); a.relu = relu; a.H = H; a.W = W; a.B = B; a.status = &status;
        xfh::rs64::strips_for(W, xfh::rs64::MAX_NSEG128, a.ns, a.ws);
        a.P = a.ws + 2; a.inv_p = 1.f / (float)a.P; a.nu = (H * a.P + 63) / 64; a.nseg = xfh::rs64::nseg_for(a.P);
        a.k = hdr[5] > 0 ? hdr[5] : xfh::rs64::runs_per_image(B * a.ns, a.nu, grid);
        const int nruns = B * a.ns * a.k, g = 4 * (nruns < grid ? nruns : grid);
        emu::launch(g, 256, xfh::rs64::lds_bytes128(a.nseg), [&] { xfh::conv_rs64_body<0, 128>(a); });
        fwrite(out.data(), 4, out.size(), stdout);
        fwrite(&status, 4, 1, stdout);
        return 0;
    }
    auto in = rd((size_t)B * 64 * H * W), w = rd(64 * 64 * 9), bias = rd(64);
    std::vector<uint16_t> wq(xfh::rs64::WQ_HALFS), wq2(4 * 2 * 3 * 64 * 8);
    xfh::pack_rs64(w.data(), wq.data());
    std::vector<float> w2, bias2;
    if (fuse) { w2 = rd(64 * 64); bias2 = rd(64); xfh::pack_rs64_1x1(w2.data(), wq2.data()); }
    std::vector<float> out((size_t)B * 64 * H * W, NAN);
    int status = 0;
    xfh::Rs64Args a{};
    a.in = in.data(); a.wq = wq.data(); a.bias = bias.data(); a.out = out.data(); a.relu = relu; a.H = H; a.W = W; a.B = B; a.status = &status;
    xfh::rs64::strips_for(W, xfh::rs64::max_nseg(fuse != 0), a.ns, a.ws);
    a.P = a.ws + 2; a.inv_p = 1.f / (float)a.P; a.nu = (H * a.P + 63) / 64; a.nseg = xfh::rs64::nseg_for(a.P);
    a.wq2 = wq2.data(); a.bias2 = fuse ? bias2.data() : nullptr; a.relu2 = hdr[7];
    a.k = hdr[5] > 0 ? hdr[5] : xfh::rs64::runs_per_image(B * a.ns, a.nu, grid);
    const int nruns = B * a.ns * a.k, g = nruns < grid ? nruns : grid;
    if (fuse == 0) emu::launch(g, 256, xfh::rs64::lds_bytes(a.nseg), [&] { xfh::conv_rs64_body<0>(a); });
    else if (fuse == 1) emu::launch(g, 256, xfh::rs64::lds_bytes(a.nseg, true), [&] { xfh::conv_rs64_body<1>(a); });
    else emu::launch(g, 256, xfh::rs64::lds_bytes(a.nseg, true), [&] { xfh::conv_rs64_body<2>(a); });
    fwrite(out.data(), 4, out.size(), stdout);
    fwrite(&status, 4, 1, stdout);
    return 0;
}
