// mnn_f16_sweep_kernel (csrc/k_match_f16.hip: the matcher's filter -- ONE pass over S^ = D1 D2^T on v_mfma_f32_32x32x16_f16, every 32 x 32 tile in both orientations, block maxima
// R / C and the row / column maxima; sliced out of the product source by tests/test_kernels_emulated.py into match_sweep_slice.hpp) on the host.
// stdin: {P, N1, N2, n1, n2, nsplit} int32 (n1 / n2: valid rows of every pair; nsplit 0: launch_match_f16's choice for 256 CUs; nsplit -1: mnn_f16_sweep2_kernel, the
// one-orientation form -- its C has 32 ceil(N1 / 1024) rows per pair, row 32 g + l = the rows {1024 g + 32 t + l} of a column), then a16 (P*N1*64), b16 (P*N2*64) as fp32
// values that are exact fp16 numbers; stdout: rowmax (P*N1), colmax (P*N2) as floats, R (P*ceil(N2/32)*N1), C (P*ceil(N1/32)*N2) (the kernel's fp16 quarter values x 4).
#include "emu.hpp"
#include <cstdio>
#include <algorithm>
#include <atomic>
using std::min;
using std::max;
#define XFH_CODE_SHIFT 0
#define XFH_EMU_NOASM(...) ((void)0)
#define XFH_KEEP_FRAGS(f) ((void)0)      // (register keep-alive of the product source: nothing to keep on the host)
typedef emu::Rsrc __amdgpu_buffer_rsrc_t;
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned o = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return o;
}
inline float emu_fetch_max_f32(float* p, float v) {
    unsigned* u = reinterpret_cast<unsigned*>(p);
    unsigned o = __atomic_load_n(u, __ATOMIC_RELAXED);
    for (;;) {
        float f; std::memcpy(&f, &o, 4);
        if (!(f < v)) return f;
        unsigned n; std::memcpy(&n, &v, 4);
        if (__atomic_compare_exchange_n(u, &o, n, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return f;
    }
}
#define __hip_atomic_fetch_max(p, v, order, scope) emu_fetch_max_f32(p, v)
// value of the wave's lane 0 (v_readfirstlane_b32 of a value only lane 0 holds: emu.hpp's readfirstlane is the identity, good for values that are uniform already)
inline int emu_bcast0(int v) {
    const int t = emu::tidx.x, w = t >> 6;
    std::memcpy(&emu::wg->opa32[t * 8], &v, 4);
    emu::wg->wave_bar[w]->arrive_and_wait();
    int r;
    std::memcpy(&r, &emu::wg->opa32[(w * 64) * 8], 4);
    emu::wg->wave_bar[w]->arrive_and_wait();
    return r;
}
// v_permlane32_swap_b32 (gfx950): the upper half of the first register changes places with the lower half of the second -- {a.lo, b.lo}, {a.hi, b.hi}
struct EmuSwap { unsigned v[2]; unsigned operator[](int i) const { return v[i]; } };
inline EmuSwap emu_permlane32_swap(unsigned a, unsigned b) {
    const float ao = emu::shfl_xor(__uint_as_float(a), 32), bo = emu::shfl_xor(__uint_as_float(b), 32);
    const bool hi = (emu::tidx.x & 32) != 0;
    return EmuSwap{{hi ? __float_as_uint(bo) : a, hi ? b : __float_as_uint(ao)}};
}
#define __builtin_amdgcn_permlane32_swap(a, b, x, y) emu_permlane32_swap(a, b)
namespace xfh {
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }      // (common.hpp)
inline unsigned float_ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }      // (common.hpp)
inline float ord_float(unsigned o) { unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o; return __uint_as_float(u); }
#include "match_sweep_slice.hpp"
}
int main() {
    int h[6];
    if (fread(h, 4, 6, stdin) != 6) return 2;
    const int P = h[0], N1 = h[1], N2 = h[2], n1v = h[3], n2v = h[4];
    std::vector<float> af((size_t)P * N1 * 64), bf((size_t)P * N2 * 64);
    if (fread(af.data(), 4, af.size(), stdin) != af.size() || fread(bf.data(), 4, bf.size(), stdin) != bf.size()) return 2;
    std::vector<_Float16> a16(af.size()), b16(bf.size());
    for (size_t i = 0; i < af.size(); ++i) a16[i] = (_Float16)af[i];
    for (size_t i = 0; i < bf.size(); ++i) b16[i] = (_Float16)bf[i];
    std::vector<int32_t> n1(P, n1v), n2(P, n2v);
    const bool one = h[5] < 0;
    const int ncb32 = (N2 + 31) / 32, nrb32 = one ? xfh::s2_c_blocks(N1) : (N1 + 31) / 32;
    std::vector<unsigned> rowmaxh((size_t)P * N1, 0u), colmaxh((size_t)P * N2, 0u);      // (zeroed by the caller: MatchWs::zeroed)
    std::vector<_Float16> R16((size_t)P * ncb32 * N1, (_Float16)NAN), C16((size_t)P * nrb32 * N2, (_Float16)NAN);      // block maxima: fp16, a quarter of the scaled product, rounded up
    const int ncc = (N2 + xfh::FT_COLS - 1) / xfh::FT_COLS;
    const int nsplit = h[5] > 0 ? h[5] : std::max(1, std::min((2 * 256 + ncc * P - 1) / (ncc * P), (N1 + 255) / 256));      // launch_match_f16, 256 CUs
    const size_t lds = sizeof(_Float16) * xfh::FT_COLS * xfh::FT_DS + sizeof(float) * 8 * xfh::FT_COLS + 64;
    if (one) {
        const int ncc2 = (N2 + xfh::S2_COLS - 1) / xfh::S2_COLS, ngq = ((N1 + xfh::S2_GROUP - 1) / xfh::S2_GROUP + xfh::S2_WAVES - 1) / xfh::S2_WAVES;
        const size_t lds2 = sizeof(_Float16) * xfh::S2_COLS * xfh::FT_DS + sizeof(_Float16) * xfh::S2_WAVES * 64 * xfh::FT_DS;
        emu::launch(ncc2 * ngq * P, 64 * xfh::S2_WAVES, lds2, [&] {
            xfh::mnn_f16_sweep2_kernel(a16.data(), (size_t)N1 * 64, b16.data(), (size_t)N2 * 64, n1.data(), n2.data(), 1, 0, N1, N2, ncc2, ngq, P, colmaxh.data(), rowmaxh.data(), R16.data(), C16.data());
        });
    } else
    emu::launch(ncc * nsplit * P, 512, lds, [&] {
        xfh::mnn_f16_sweep_kernel(a16.data(), (size_t)N1 * 64, b16.data(), (size_t)N2 * 64, n1.data(), n2.data(), 1, 0, N1, N2, ncc, nsplit, P, colmaxh.data(), rowmaxh.data(), R16.data(), C16.data());
    });
    std::vector<float> rm(rowmaxh.size()), cm(colmaxh.size());
    for (size_t i = 0; i < rm.size(); ++i) rm[i] = rowmaxh[i] ? xfh::ord_float(rowmaxh[i]) : NAN;      // (0 = never written)
    for (size_t i = 0; i < cm.size(); ++i) cm[i] = colmaxh[i] ? xfh::ord_float(colmaxh[i]) : NAN;
    fwrite(rm.data(), 4, rm.size(), stdout); fwrite(cm.data(), 4, cm.size(), stdout);
    std::vector<float> R(R16.size()), C(C16.size());      // (written out in the product's units: x 4)
    for (size_t i = 0; i < R.size(); ++i) R[i] = 4.f * (float)R16[i];
    for (size_t i = 0; i < C.size(); ++i) C[i] = 4.f * (float)C16[i];
    fwrite(R.data(), 4, R.size(), stdout); fwrite(C.data(), 4, C.size(), stdout);
    return 0;
}
