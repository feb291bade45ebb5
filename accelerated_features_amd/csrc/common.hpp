// Shared device/host helpers for libxfeat_hip (gfx950 only: 64-lane wavefronts assumed).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace xfh {

constexpr int WAVE = 64;

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- wavefront reductions (64 lanes) ---------------------------------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline unsigned long long shfl_xor_u64(unsigned long long v, int o) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor(lo, o, 64);
    hi = __shfl_xor(hi, o, 64);
    return ((unsigned long long)hi << 32) | lo;
}
// Value held by the lane 32 positions away (lane ^ 32), without the LDS crossbar: v_permlane32_swap_b32 (gfx950) swaps
// the upper half of one register with the lower half of another in a single VALU op; __shfl_xor(v, 32) costs two
// address ops, a ds_bpermute and an lgkmcnt wait.
__device__ inline unsigned xhalf_u32(unsigned v) {
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);      // r[0] = {v.lo, v.lo}, r[1] = {v.hi, v.hi}
    return (threadIdx.x & 32) ? r[0] : r[1];
}
__device__ inline float xhalf(float v) { return __uint_as_float(xhalf_u32(__float_as_uint(v))); }
__device__ inline unsigned long long xhalf_u64(unsigned long long v) {
    return ((unsigned long long)xhalf_u32((unsigned)(v >> 32)) << 32) | xhalf_u32((unsigned)v);
}
__device__ inline unsigned long long u64_max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }

// Monotone map float -> uint32: a < b  <=>  ord(a) < ord(b)   (NaN not expected on this path).
__device__ inline unsigned float_ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float ord_float(unsigned o) {
    unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}

// Barrier that covers global->LDS DMA (global_load_lds) issued by ANY wave of the workgroup.
// The explicit asm wait is mandatory: hipcc (ROCm 7.2) was observed to omit the vmcnt(0) in
// front of s_barrier on a loop back-edge when the DMA sits in exec-masked blocks (the 24->24
// conv instantiation: ~7 % of launches consumed a staging buffer before it had landed).
// Inline asm is invisible to the waitcnt pass, so it cannot be dropped.
__device__ inline void lds_dma_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// Debug hooks at the entry of every matrix-core kernel.  XFH_CODE_SHIFT (build.py --shift N -> libxfeat_hip_shiftN.so): the kernel body moved by 4 N bytes
// against the 64-byte instruction-cache lines; cold (xfh_debug_cold_start): the workgroup starts on an invalidated instruction cache -- together the
// code-position scan of DESIGN 9.0 (tools/shift_scan.sh) for the whole library.  The production build has N = 0 and cold = 0: one scalar compare.
#ifndef XFH_CODE_SHIFT
#define XFH_CODE_SHIFT 0
#endif
template <int N> __device__ inline void code_shift() {
    if constexpr (N > 0) { asm volatile("s_nop 0"); code_shift<N - 1>(); }
}
__device__ inline void kernel_entry_hooks(int cold) {
    code_shift<XFH_CODE_SHIFT>();
    if (cold) asm volatile("s_icache_inv\n\ts_nop 7\n\ts_nop 7");
}

// Raise a kernel's dynamic-LDS limit once per device.  `mask` is a static of the call site (bit d = done on device d): function
// attributes are per device, so a process that drives several GPUs must set them on each (a per-process "done" flag would leave
// every device but the first at the 64 KB default).  The mask is an atomic: threads of one process (one handle per stream) race on
// it by design -- at worst both make the idempotent call -- and the only process-wide mutable state of the library stays well defined.
using AttrMask = std::atomic<unsigned>;
inline void set_max_dynamic_lds(const void* fn, int bytes, AttrMask& mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 32 && ((mask.load(std::memory_order_acquire) >> dev) & 1u)) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (dev >= 0 && dev < 32) mask.fetch_or(1u << dev, std::memory_order_release);
}

// compute units of the current device (256 on MI355X); persistent kernels size their grids with it
inline int num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// XCD-aware work mapping (MI355X: 8 XCDs, private L2s; workgroup b runs on XCD b % 8).
// A 1-D grid of n_groups_padded * per_group workgroups is remapped so that all `per_group`
// workgroups of a group (an image, a descriptor pair) run on ONE XCD and share its L2:
// group g lives on XCD g % 8.  Placement only affects speed, never results.
// Returns false for the padding workgroups (group >= n_groups).
// The remap is used only when n_groups is a multiple of 8 (otherwise XCDs would be loaded
// unevenly, or idle for tiny batches) -- both sides of the launch use the same rule.
__host__ __device__ inline bool xcd_swizzled(int n_groups) { return n_groups >= 8 && (n_groups & 7) == 0; }
__device__ inline bool xcd_group_map(int id, int per_group, int n_groups, int& group, int& item) {
    if (!xcd_swizzled(n_groups)) {
        group = id / per_group;
        item = id - group * per_group;
        return group < n_groups;
    }
    const int xcd = id & 7, slot = id >> 3;
    group = (slot / per_group) * 8 + xcd;
    item = slot % per_group;
    return group < n_groups;
}
__host__ inline int xcd_grid_size(int per_group, int n_groups) { return per_group * n_groups; }

}  // namespace xfh
