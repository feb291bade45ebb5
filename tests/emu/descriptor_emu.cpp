// descriptor_kernel (csrc/k_detect.hip: F.normalize + bicubic sampling + F.normalize of the selected key-points, modules/xfeat.py:70,90-103, and the epilogue of the top-k:
// key-points, scores, n_valid; sliced out of the product source by tests/test_descriptor_emulated.py into descriptor_slice.hpp) on the host.  The kernel's lane exchanges are
// DPP operands within rows of 16 lanes (a key-point's lanes), and rows past an image's list leave as a whole: emu.hpp's update_dpp synchronises per row.
// stdin: {B, H, W, cap, top_k} int32, rw, rh fp32, then feats (B, H/8 * W/8, 64) fp32, inv (B, H/8 * W/8) fp32, cand (B, cap) u32, skeys (B, top_k) u64, nsel (B) i32;
// stdout: kpts (B, top_k, 2), scores (B, top_k), desc (B, top_k, 64) fp32, desc16 (B, top_k, 64) u16, n_valid (B) i32.
#include "emu.hpp"
#include <cstdio>
struct ushort4 { unsigned short x, y, z, w; };
inline ushort4 make_ushort4(unsigned short x, unsigned short y, unsigned short z, unsigned short w) { return ushort4{x, y, z, w}; }
inline float __fmaf_rn(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
namespace xfh {
inline float ord_float(unsigned o) { const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o; float f; std::memcpy(&f, &u, 4); return f; }      // (common.hpp)
#include "descriptor_slice.hpp"
}
template <typename T> static std::vector<T> rd(size_t n) {
    std::vector<T> v(n);
    if (fread(v.data(), sizeof(T), n, stdin) != n) { fprintf(stderr, "short input\n"); exit(2); }
    return v;
}
int main() {
    int h[5]; float r[2];
    if (fread(h, 4, 5, stdin) != 5 || fread(r, 4, 2, stdin) != 2) return 2;
    const int B = h[0], H = h[1], W = h[2], cap = h[3], top_k = h[4], npix = (H / 8) * (W / 8);
    auto feats = rd<float>((size_t)B * npix * 64), inv = rd<float>((size_t)B * npix);
    auto cand = rd<unsigned>((size_t)B * cap);
    auto skeys = rd<unsigned long long>((size_t)B * top_k);
    auto nsel = rd<int>(B);
    std::vector<float> kpts((size_t)B * top_k * 2, NAN), scores((size_t)B * top_k, NAN), desc((size_t)B * top_k * 64, NAN);
    std::vector<uint16_t> d16((size_t)B * top_k * 64, 0xffff);
    std::vector<int32_t> nv(B, 0);
    const int bpi = (top_k + 15) / 16;
    emu::launch(B * bpi, 256, 0, [&] { xfh::descriptor_kernel(feats.data(), inv.data(), cand.data(), skeys.data(), nsel.data(), H, W, cap, top_k, B, bpi, r[0], r[1], kpts.data(), scores.data(), nv.data(), desc.data(), d16.data()); });
    fwrite(kpts.data(), 4, kpts.size(), stdout); fwrite(scores.data(), 4, scores.size(), stdout); fwrite(desc.data(), 4, desc.size(), stdout);
    fwrite(d16.data(), 2, d16.size(), stdout); fwrite(nv.data(), 4, nv.size(), stdout);
    return 0;
}
