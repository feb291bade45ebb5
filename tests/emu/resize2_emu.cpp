// resize2_gray_stats_kernel (four-byte gathers) and resize2_gray_stats_lds_kernel (the tile's input region staged in LDS one tile ahead, tap tables) + gray_coef_kernel
// (csrc/k_preproc.hip: the dual-scale dense path's two bilinear resizes fused into the gray kernel, modules/xfeat.py:379-381, 234-238, modules/model.py:135-136; sliced out of
// the product source by tests/test_kernels_emulated.py into resize2_slice.hpp) on the host.  stdin: {B, C, Hin, Win, Hm, Wm, Ho, Wo} int32, {s1h, s1w, s2h, s2w} fp32, then
// img (B*C*Hin*Win) fp32; stdout: gray of the gather form (B*Ho*Wo), gray of the staged form, coef (B*2) of each.
#include "emu.hpp"
#include <cstdio>
#include <algorithm>
using std::min;
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline float fminf_(float a, float b) { return a < b ? a : b; }
namespace xfh {
constexpr int GS_CHUNKS = 64;                                   // (kernels.hpp)
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }   // (common.hpp)
inline double wave_sum(double v) {                              // (common.hpp: the same butterfly)
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
#include "resize2_slice.hpp"
}
int main() {
    int h[8]; float sc[4];
    if (fread(h, 4, 8, stdin) != 8 || fread(sc, 4, 4, stdin) != 4) return 2;
    const int B = h[0], C = h[1], Hin = h[2], Win = h[3], Hm = h[4], Wm = h[5], Ho = h[6], Wo = h[7];
    const float s1h = sc[0], s1w = sc[1], s2h = sc[2], s2w = sc[3];
    std::vector<float> img((size_t)B * C * Hin * Win);
    if (fread(img.data(), 4, img.size(), stdin) != img.size()) return 2;
    std::vector<float> g0((size_t)B * Ho * Wo, NAN), g1((size_t)B * Ho * Wo, NAN), c0(2 * B, NAN), c1(2 * B, NAN);
    std::vector<double> part((size_t)B * xfh::GS_CHUNKS * 2, NAN);
    // launch_gray_norm_resized's sizes
    const int rh = (int)(s2h * (xfh::R2_TH - 1)) + 3, rw = (int)(s2w * (xfh::R2_TW - 1)) + 3;
    const size_t lds = (size_t)C * rh * rw * sizeof(float);
    const int in_rows = (int)(s1h * (rh - 1)) + 3, in_w = ((int)(s1w * (rw - 1)) + 3 + 3 + 3) / 4 * 4, mid_cap = (rh * rw + 3) / 4 * 4;
    const size_t lds2 = ((size_t)4 * xfh::R2_NTAB + (size_t)C * mid_cap + (size_t)C * in_rows * in_w) * sizeof(float);
    if (Win % 4 || in_w > 128 || in_rows > 32 || C > 3) { fprintf(stderr, "the staged form does not take this shape\n"); return 3; }
    emu::launch(xfh::GS_CHUNKS * B, 256, lds + 64, [&] { xfh::resize2_gray_stats_kernel(img.data(), C, Hin, Win, Hm, Wm, s1h, s1w, Ho, Wo, s2h, s2w, part.data(), g0.data()); });
    emu::launch(B, 64, 64, [&] { xfh::gray_coef_kernel(part.data(), Ho * Wo, 1e-5f, c0.data()); });
    std::fill(part.begin(), part.end(), NAN);
    emu::launch(xfh::GS_CHUNKS * B, 256, lds2 + 64, [&] { xfh::resize2_gray_stats_lds_kernel(img.data(), C, Hin, Win, Hm, Wm, s1h, s1w, Ho, Wo, s2h, s2w, part.data(), g1.data(), mid_cap, in_rows, in_w); });
    emu::launch(B, 64, 64, [&] { xfh::gray_coef_kernel(part.data(), Ho * Wo, 1e-5f, c1.data()); });
    fwrite(g0.data(), 4, g0.size(), stdout);
    fwrite(g1.data(), 4, g1.size(), stdout);
    fwrite(c0.data(), 4, c0.size(), stdout);
    fwrite(c1.data(), 4, c1.size(), stdout);
    return 0;
}
