// pyramid53_kernel<4> (csrc/k_preproc.hip: block5.3's 1x1 convolution fused into the pyramid sum, modules/model.py:78,146-148; sliced out of the product source by
// tests/test_kernels_emulated.py into pyramid_slice.hpp) on the host.  stdin: {B, H3, W3, H4, W4, H5, W5, relu | 2 generic taps} int32, then x3 (B,64,H3,W3), x4 (B,64,H4,W4),
// y5 (B,128,H5,W5), the weights as [128][64] and the 64 biases, fp32; stdout: out (B,64,H3,W3).
#include "emu.hpp"
#include <cstdio>
#include <algorithm>
using std::min;
using std::max;
inline int __mul24(int a, int b) { return a * b; }      // (v_mul_i32_i24: the kernel keeps both factors below 2^23)
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
    const int B = h[0], H3 = h[1], W3 = h[2], H4 = h[3], W4 = h[4], H5 = h[5], W5 = h[6];
    auto x3 = rd((size_t)B * 64 * H3 * W3), x4 = rd((size_t)B * 64 * H4 * W4), y5 = rd((size_t)B * 128 * H5 * W5), w = rd(128 * 64), bias = rd(64);
    std::vector<float> out(x3.size(), NAN);
    constexpr int CG = 4;
    const size_t n4s = std::max((size_t)CG * H4 * W4, (size_t)4 * CG * H5 * W5);
    const size_t lds = (((n4s + 3) & ~(size_t)3) + (((size_t)CG * H5 * W5 + 3) & ~(size_t)3)) * sizeof(float) + 2 * (size_t)(W3 + H3) * sizeof(xfh::PyrCoef) + 2 * (size_t)W3 * 8;      // launch_pyramid53's own sizing
    if ((W3 & 3) || lds > 64 * 1024) { fprintf(stderr, "not this kernel's case\n"); return 3; }
    emu::launch(B * (64 / CG), 256, lds, [&] { xfh::pyramid53_kernel<CG>(x3.data(), x4.data(), y5.data(), w.data(), bias.data(), h[7] & 1, out.data(), H3, W3, H4, W4, H5, W5, (h[7] >> 1) & 1); });
    fwrite(out.data(), 4, out.size(), stdout);
    return 0;
}
