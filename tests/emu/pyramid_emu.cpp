// pyramid_sum_kernel<USE_LDS> (csrc/k_preproc.hip: x3 + up(x4) + up(x5), modules/model.py:146-148; sliced out of the product source by tests/test_kernels_emulated.py into
// pyramid_slice.hpp) on the host.  stdin: {planes, H3, W3, H4, W4, H5, W5, use_lds} int32, then x3, x4, x5 as fp32 planes; stdout: out (planes*H3*W3).
#include "emu.hpp"
#include <cstdio>
#include <algorithm>
using std::min;
using std::max;
inline int __mul24(int a, int b) { return a * b; }
namespace xfh {
#include "pyramid_slice.hpp"
}
static std::vector<float> rd(size_t n) {
    std::vector<float> v(n);
    if (fread(v.data(), 4, n, stdin) != n) { fprintf(stderr, "short input\n"); exit(2); }
    return v;
}
int main() {
    int h[8];
    if (fread(h, 4, 8, stdin) != 8) return 2;
    const int planes = h[0], H3 = h[1], W3 = h[2], H4 = h[3], W4 = h[4], H5 = h[5], W5 = h[6];
    auto x3 = rd((size_t)planes * H3 * W3), x4 = rd((size_t)planes * H4 * W4), x5 = rd((size_t)planes * H5 * W5);
    std::vector<float> out(x3.size(), NAN);
    const size_t lds = ((((size_t)H4 * W4 + (size_t)H5 * W5 + 3) & ~(size_t)3) * sizeof(float)) + 2 * (size_t)(W3 + H3) * sizeof(xfh::PyrCoef);      // launch_pyramid_sum's own sizing
    if (h[7] && lds <= 64 * 1024) emu::launch(planes, 256, lds, [&] { xfh::pyramid_sum_kernel<true>(x3.data(), x4.data(), x5.data(), out.data(), H3, W3, H4, W4, H5, W5); });
    else emu::launch(planes, 256, 16, [&] { xfh::pyramid_sum_kernel<false>(x3.data(), x4.data(), x5.data(), out.data(), H3, W3, H4, W4, H5, W5); });
    fwrite(out.data(), 4, out.size(), stdout);
    return 0;
}
