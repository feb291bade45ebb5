// Host emulation of the few GPU facilities a kernel body of csrc/ needs (tests/emu/*.cpp): one host thread per work-item, a workgroup at a time.
// LDS is a buffer, __syncthreads a barrier over the workgroup's threads, the LDS-DMA a copy, v_mfma_f32_16x16x32_f16 an exchange among the 64 threads of a wave
// (exact fp16 products, the accumulator rounded to fp32 once per instruction: what tests/test_block1_fx_model.py assumes of the instruction).
// Not a model of timing or of memory ordering: it checks index arithmetic, tile layouts and the barrier structure (a missing barrier usually shows as a wrong
// result here too, since the host threads run at very different paces).
#pragma once
#define XFH_HOST_EMU 1
#include <atomic>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <thread>
#include <type_traits>
#include <vector>

#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)

struct emu_dim3 { unsigned x = 0, y = 0, z = 0; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
using std::max;
using std::min;

namespace emu {
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
struct Workgroup {
    int nthreads;
    std::vector<unsigned char> lds;
    std::barrier<> bar;
    std::vector<std::unique_ptr<std::barrier<>>> wave_bar;
    std::vector<std::unique_ptr<std::barrier<>>> row_bar;      // one per 16 lanes (a DPP row): for kernels whose 16-lane groups leave early as a whole (descriptor_kernel)
    std::vector<h8> opa, opb;      // [thread]
    std::vector<float> opa32, opb32;      // [thread][8]
    Workgroup(int n, size_t lds_bytes) : nthreads(n), lds(lds_bytes + 64, 0xff), bar(n), opa(n), opb(n), opa32(8 * n), opb32(8 * n) {      // (LDS starts as NaN patterns: nothing may rely on zeros)
        for (int w = 0; w < n / 64; ++w) wave_bar.emplace_back(new std::barrier<>(64));
        for (int r = 0; r < n / 16; ++r) row_bar.emplace_back(new std::barrier<>(16));
    }
    unsigned char* lds_base() { return reinterpret_cast<unsigned char*>((reinterpret_cast<uintptr_t>(lds.data()) + 63) & ~uintptr_t(63)); }
};
inline thread_local Workgroup* wg = nullptr;
inline thread_local emu_dim3 tidx, bidx;
inline emu_dim3 gdim;

inline f4 mfma16(h8 a, h8 b, f4 c) {
    const int t = tidx.x, w = t >> 6, l = t & 63;
    wg->opa[t] = a; wg->opb[t] = b;
    wg->wave_bar[w]->arrive_and_wait();
    f4 d = c;
    const int n = l & 15;                       // this lane's column
    for (int j = 0; j < 4; ++j) {
        const int m = 4 * (l >> 4) + j;         // row
        double s = 0;
        for (int kb = 0; kb < 4; ++kb)
            for (int i = 0; i < 8; ++i) s += (double)(float)wg->opa[w * 64 + 16 * kb + m][i] * (double)(float)wg->opb[w * 64 + 16 * kb + n][i];
        d[j] = (float)((double)c[j] + s);
    }
    wg->wave_bar[w]->arrive_and_wait();
    return d;
}
// v_mfma_f32_32x32x16_{f16,bf16}: lane l holds A row l & 31 / B column l & 31, K values 8 (l >> 5) .. + 7; D: lane (column l & 31, half l >> 5) holds rows
// (r & 3) + 8 (r >> 2) + 4 half, r = 0 .. 15
typedef float f16v __attribute__((ext_vector_type(16)));
template <typename V8>
inline f16v mfma32(V8 a, V8 b, f16v c) {
    const int t = tidx.x, w = t >> 6, l = t & 63;
    h8 af, bf;      // (as fp32-exact carriers: both fp16 and bf16 convert to float exactly; kept as floats below)
    static thread_local float fa[8], fb[8];
    for (int i = 0; i < 8; ++i) { fa[i] = (float)a[i]; fb[i] = (float)b[i]; }
    std::memcpy(&wg->opa32[t * 8], fa, 32); std::memcpy(&wg->opb32[t * 8], fb, 32);
    (void)af; (void)bf;
    wg->wave_bar[w]->arrive_and_wait();
    f16v d = c;
    const int n = l & 31, half = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
        double s = 0;
        for (int kb = 0; kb < 2; ++kb)
            for (int i = 0; i < 8; ++i) s += (double)wg->opa32[(w * 64 + 32 * kb + m) * 8 + i] * (double)wg->opb32[(w * 64 + 32 * kb + n) * 8 + i];
        d[r] = (float)((double)c[r] + s);
    }
    wg->wave_bar[w]->arrive_and_wait();
    return d;
}
// v_mfma_f32_32x32x2_f32: lane l holds A[row l & 31][k = l >> 5] and B[k = l >> 5][column l & 31]; D as above
inline f16v mfma32x2(float a, float b, f16v c) {
    const int t = tidx.x, w = t >> 6, l = t & 63;
    wg->opa32[t * 8] = a; wg->opb32[t * 8] = b;
    wg->wave_bar[w]->arrive_and_wait();
    f16v d = c;
    const int n = l & 31, half = l >> 5;
    for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
        double s = 0;
        for (int k = 0; k < 2; ++k) s += (double)wg->opa32[(w * 64 + 32 * k + m) * 8] * (double)wg->opb32[(w * 64 + 32 * k + n) * 8];
        d[r] = (float)((double)c[r] + s);
    }
    wg->wave_bar[w]->arrive_and_wait();
    return d;
}
// v_mov_b32_dpp within a ROW of 16 lanes (all that the kernels emulated here use): quad_perm (ctrl < 0x100), row_mirror (0x140), row_half_mirror (0x141), row_newbcast:n
// (0x150 + n).  Every source lane of these is in the lane's own row and the rows' lanes reach the instruction together, so the exchange synchronises the row, not the wave --
// a row whose 16 lanes have all returned (descriptor_kernel: rows past the list) is simply absent.  row_mask / bank_mask 0xf, bound_ctrl irrelevant (no invalid source).
inline unsigned update_dpp(unsigned, unsigned src, int ctrl, int row_mask, int bank_mask, bool) {
    const int t = tidx.x, l = t & 15, base = t & ~15;
    if (row_mask != 0xf || bank_mask != 0xf) std::abort();
    std::memcpy(&wg->opb32[t * 8], &src, 4);
    wg->row_bar[t >> 4]->arrive_and_wait();
    int sl;
    if (ctrl < 0x100) sl = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
    else if (ctrl == 0x140) sl = 15 - l;
    else if (ctrl == 0x141) sl = (l & 8) | (7 - (l & 7));
    else if (ctrl >= 0x150 && ctrl < 0x160) sl = ctrl - 0x150;
    else std::abort();
    unsigned r;
    std::memcpy(&r, &wg->opb32[(base + sl) * 8], 4);
    wg->row_bar[t >> 4]->arrive_and_wait();
    return r;
}
// value of lane ^ mask of the same wave
inline float shfl_xor(float v, int mask) {
    const int t = tidx.x, w = t >> 6, l = t & 63;
    wg->opa32[t * 8] = v;
    wg->wave_bar[w]->arrive_and_wait();
    const float r = wg->opa32[(w * 64 + (l ^ mask)) * 8];
    wg->wave_bar[w]->arrive_and_wait();
    return r;
}
inline double shfl_xor(double v, int mask) {      // (a thread's exchange slot is 32 bytes wide)
    const int t = tidx.x, w = t >> 6, l = t & 63;
    std::memcpy(&wg->opa32[t * 8], &v, 8);
    wg->wave_bar[w]->arrive_and_wait();
    double r;
    std::memcpy(&r, &wg->opa32[(w * 64 + (l ^ mask)) * 8], 8);
    wg->wave_bar[w]->arrive_and_wait();
    return r;
}
inline void dma(const void* g, void* l_base, int size) { std::memcpy(static_cast<unsigned char*>(l_base) + size * (tidx.x & 63), g, size); }      // (M0 base + lane x size)
struct Rsrc { unsigned char* base; unsigned bytes; };
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
// raw buffer load of 16 bytes: the range check covers the lane's offset (voff), not the scalar offset; out of range reads zeros
inline u4 buffer_load_b128(Rsrc rs, unsigned voff, unsigned soff) {
    u4 r = {0u, 0u, 0u, 0u};
    if ((uint64_t)voff + 16 <= rs.bytes) std::memcpy(&r, rs.base + voff + soff, 16);
    return r;
}
// buffer_load_dwordx4 ... lds with a hand-built resource word (base address in .x/.y, bytes in .z): 16 bytes per lane to the LDS offset m0 + 16 lane
// EMU_DEFER_DMA: the copy happens as LATE as the kernel's own waits allow -- a thread's requests queue up and s_waitcnt vmcnt(n) (XFH_WAIT_VMCNT) completes all but the n
// youngest.  Without it the copy happens at issue, the EARLIEST the hardware could deliver it.  A ring that is right under both orders reads no slot before its
// wait and overwrites none that is still being read.  (Only for kernels whose vector-memory queue holds nothing but these requests wherever they use a partial count.)
struct PendingDma { unsigned char* dst; const unsigned char* src; };
inline thread_local std::vector<PendingDma> dma_queue;
inline void dma_wait(size_t keep) {
    const size_t n = dma_queue.size() > keep ? dma_queue.size() - keep : 0;
    for (size_t i = 0; i < n; ++i) { if (dma_queue[i].src) std::memcpy(dma_queue[i].dst, dma_queue[i].src, 16); else std::memset(dma_queue[i].dst, 0, 16); }
    dma_queue.erase(dma_queue.begin(), dma_queue.begin() + n);
}
inline void dma_b128_to_lds(unsigned m0v, unsigned voff, i4 rs, unsigned soff) {
    const unsigned char* base = reinterpret_cast<const unsigned char*>(((uint64_t)((unsigned)rs.y & 0xffffu) << 32) | (unsigned)rs.x);
    unsigned char* dst = wg->lds_base() + m0v + 16 * (tidx.x & 63);
#ifdef EMU_DEFER_DMA
    dma_queue.push_back({dst, (uint64_t)voff + 16 <= (unsigned)rs.z ? base + voff + soff : nullptr});
#else
    if ((uint64_t)voff + 16 <= (unsigned)rs.z) std::memcpy(dst, base + voff + soff, 16); else std::memset(dst, 0, 16);
#endif
}
// buffer_load_dword ... lds: 4 bytes per lane to the LDS offset m0 + 4 lane; a lane whose offset is out of the resource's range (the kernels use 0x80000000) writes zero
inline void dma_b32_to_lds(unsigned m0v, unsigned voff, i4 rs, unsigned soff) {
    const unsigned char* base = reinterpret_cast<const unsigned char*>(((uint64_t)((unsigned)rs.y & 0xffffu) << 32) | (unsigned)rs.x);
    unsigned char* dst = wg->lds_base() + m0v + 4 * (tidx.x & 63);
    if ((uint64_t)voff + 4 <= (unsigned)rs.z) std::memcpy(dst, base + voff + soff, 4); else std::memset(dst, 0, 4);
}
inline unsigned buffer_load_b32(Rsrc rs, unsigned voff, unsigned soff) {
    unsigned r = 0;
    if ((uint64_t)voff + 4 <= rs.bytes) std::memcpy(&r, rs.base + voff + soff, 4);
    return r;
}
inline float med3(float a, float b, float c) { return std::max(std::min(a, b), std::min(std::max(a, b), c)); }
inline unsigned perm(unsigned hi, unsigned lo, unsigned sel) {
    const uint64_t v = ((uint64_t)hi << 32) | lo;
    unsigned r = 0;
    for (int i = 0; i < 4; ++i) r |= (unsigned)((v >> (8 * ((sel >> (8 * i)) & 7))) & 0xff) << (8 * i);
    return r;
}

// run `fn()` as every work-item of `grid` workgroups of `nthreads`, one workgroup at a time
template <typename F>
void launch(int grid, int nthreads, size_t lds_bytes, F fn) {
    Workgroup w(nthreads, lds_bytes);
    gdim.x = grid;
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t] {
            wg = &w;
            tidx.x = t;
            for (int b = 0; b < grid; ++b) {
                bidx.x = b;
                fn();
                w.bar.arrive_and_wait();        // the next workgroup reuses the LDS
            }
        });
    for (auto& t : th) t.join();
}
}  // namespace emu

#define threadIdx emu::tidx
#define blockIdx emu::bidx
#define gridDim emu::gdim
inline void __syncthreads() { emu::wg->bar.arrive_and_wait(); }
inline int atomicOr(int* p, int v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
#define XFH_DYN_LDS(name) float* name = reinterpret_cast<float*>(emu::wg->lds_base())
#define XFH_DYN_LDS_BYTES(name) unsigned char* name = emu::wg->lds_base()
#define XFH_LDS_VOLATILE(T) volatile T
#define XFH_NOP16_2(a, b) ((void)0)
#define XFH_NOP16() ((void)0)
#define XFH_NOP16_4(a, b, c, d) ((void)0)
#define XFH_NOP32_2(a, b) ((void)0)
#ifdef EMU_DEFER_DMA
#define XFH_WAIT_VMCNT0() emu::dma_wait(0)
#define XFH_WAIT_VMCNT(n) emu::dma_wait(n)
#else
#define XFH_WAIT_VMCNT0() ((void)0)
#define XFH_WAIT_VMCNT(n) ((void)0)
#endif
#define XFH_WAIT_LGKMCNT0() ((void)0)
#define XFH_DMA_PROTOCOL_EMULATED() ((void)0)
#define XFH_LDS_ADDR(p, base) ((unsigned)((p) - (base)))
#define XFH_DMA_B128_TO_LDS(m0v, voff, rsrc, soff) emu::dma_b128_to_lds(m0v, voff, rsrc, soff)
#define XFH_NOP16_3(a, b, c) ((void)0)
#define XFH_PIN(x) ((void)0)
#define XFH_AGPR(x) ((void)0)
#define XFH_VGPR(x) ((void)0)
#define XFH_AGPR_ACC(c0, c1) ((void)0)
#define XFH_AGPR_TAP(h, l, c0, c1) ((void)0)
#define XFH_AGPR_TAP2(h, l, h1, l1, c0, c1) ((void)0)
#define XFH_LDS_BARRIER() __syncthreads()
#define XFH_SCHED_FENCE() ((void)0)
#define XFH_WAVE_SYNC() emu::wg->wave_bar[emu::tidx.x >> 6]->arrive_and_wait()      /* lanes of a wave run in lock-step on the GPU; here they are threads */
typedef const void* xfh_gptr_t;
typedef void* xfh_lptr_t;
typedef emu::Rsrc __amdgpu_buffer_rsrc_t;
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) emu::update_dpp(old, src, ctrl, rm, bm, bc)
#define __builtin_amdgcn_fmed3f(a, b, c) emu::med3(a, b, c)
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
#define __builtin_amdgcn_perm(hi, lo, sel) emu::perm(hi, lo, sel)
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) emu::mfma16(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) emu::mfma32(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) emu::mfma32(a, b, c)
#define __builtin_amdgcn_s_memtime() 0ll
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu::mfma32x2(a, b, c)
#define __builtin_amdgcn_sched_group_barrier(mask, n, id) ((void)0)
#define __shfl_xor(v, mask, width) emu::shfl_xor(v, mask)
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) emu::dma(g, l, size)
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, bytes, flags) emu::Rsrc{reinterpret_cast<unsigned char*>(p), (unsigned)(bytes)}
#define __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, aux) emu::buffer_load_b128(rs, (unsigned)(voff), (unsigned)(soff))
#define __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, aux) emu::buffer_load_b32(rs, (unsigned)(voff), (unsigned)(soff))
#define __builtin_amdgcn_raw_buffer_store_b128(val, rs, voff, soff, aux)                                                              \
    do { const unsigned vo_ = (unsigned)(voff); if ((uint64_t)vo_ + 16 <= (rs).bytes) { const auto v_ = (val); std::memcpy((rs).base + vo_ + (unsigned)(soff), &v_, 16); } } while (0)
#define __builtin_amdgcn_raw_buffer_store_b64(val, rs, voff, soff, aux)                                                               \
    do { const unsigned vo_ = (unsigned)(voff); if ((uint64_t)vo_ + 8 <= (rs).bytes) { const auto v_ = (val); std::memcpy((rs).base + vo_ + (unsigned)(soff), &v_, 8); } } while (0)
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_raw_buffer_store_b16(val, rs, voff, soff, aux)                                                               \
    do { const unsigned vo_ = (unsigned)(voff); if ((uint64_t)vo_ + 2 <= (rs).bytes) { const short v_ = (val); std::memcpy((rs).base + vo_ + (unsigned)(soff), &v_, 2); } } while (0)
#define __builtin_amdgcn_raw_buffer_store_b32(val, rs, voff, soff, aux)                                                               \
    do { const unsigned vo_ = (unsigned)(voff); if ((uint64_t)vo_ + 4 <= (rs).bytes) { const unsigned v_ = (val); std::memcpy((rs).base + vo_ + (unsigned)(soff), &v_, 4); } } while (0)

namespace xfh {
inline float xhalf(float v) { return emu::shfl_xor(v, 32); }      // (common.hpp: v_permlane32_swap)
inline void kernel_entry_hooks(int) {}
inline void lds_dma_barrier() { __syncthreads(); }
inline bool xcd_swizzled(int n_groups) { return n_groups >= 8 && (n_groups & 7) == 0; }
inline bool xcd_group_map(int id, int per_group, int n_groups, int& group, int& item) {      // (common.hpp)
    if (!xcd_swizzled(n_groups)) { group = id / per_group; item = id - group * per_group; return group < n_groups; }
    const int xcd = id & 7, slot = id >> 3;
    group = (slot / per_group) * 8 + xcd; item = slot % per_group;
    return group < n_groups;
}
}  // namespace xfh
