// gray_stats_kernel<CT> + gray_coef_kernel (csrc/k_preproc.hip: channel mean and InstanceNorm2d(1) as a per-image affine map, modules/model.py:135-136; sliced out of the
// product source by tests/test_kernels_emulated.py into gray_slice.hpp) on the host.  stdin: {B, C, H, W} int32, then img (B*C*H*W) fp32; stdout: gray (B*H*W), coef (B*2).
#include "emu.hpp"
#include <cstdio>
#include <algorithm>
using std::min;
namespace xfh {
constexpr int GS_CHUNKS = 64;                                   // (kernels.hpp)
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }   // (common.hpp)
inline double wave_sum(double v) {                              // (common.hpp: the same butterfly)
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
#include "gray_slice.hpp"
}
int main() {
    int h[4];
    if (fread(h, 4, 4, stdin) != 4) return 2;
    const int B = h[0], C = h[1], H = h[2], W = h[3], HW = H * W;
    std::vector<float> img((size_t)B * C * HW), gray((size_t)B * HW, NAN), coef(2 * B, NAN);
    if (fread(img.data(), 4, img.size(), stdin) != img.size()) return 2;
    std::vector<double> part((size_t)B * xfh::GS_CHUNKS * 2, NAN);
    auto go = [&](auto fn) { emu::launch(xfh::GS_CHUNKS * B, 256, 64, fn); };      // launch_gray_norm's choice of the instantiation
    if (C == 3) go([&] { xfh::gray_stats_kernel<3>(img.data(), C, HW, part.data(), gray.data()); });
    else if (C == 1) go([&] { xfh::gray_stats_kernel<1>(img.data(), C, HW, part.data(), gray.data()); });
    else go([&] { xfh::gray_stats_kernel<0>(img.data(), C, HW, part.data(), gray.data()); });
    emu::launch(B, 64, 64, [&] { xfh::gray_coef_kernel(part.data(), HW, 1e-5f, coef.data()); });
    fwrite(gray.data(), 4, gray.size(), stdout);
    fwrite(coef.data(), 4, coef.size(), stdout);
    return 0;
}
