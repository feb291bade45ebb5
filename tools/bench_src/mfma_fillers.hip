// What does an instruction placed between a wave's OWN v_mfma_f32_32x32x16_f16 cost when that wave is alone on its SIMD (conv_rs64_kernel's situation)?
//   hipcc -O2 --offload-arch=gfx950 tools/bench_src/mfma_fillers.hip -o gpurun_probe/mfma_fillers ; gpurun_probe/mfma_fillers
// One workgroup of four waves per CU, a stream of MFMAs alternating between two accumulators (everything in inline asm: program order = issue order), NF fillers of one
// KIND behind every MFMA.  Prints shader cycles (s_memtime) per MFMA.  The matrix pipe's floor is 32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int NF>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 c0, c1;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
    f16x8 a, b;
    for (int r = 0; r < 8; ++r) { a[r] = (_Float16)(0.001f * lane); b[r] = (_Float16)(0.002f * r); }
    unsigned v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane + i;
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = 0.5f * lane + i;
    unsigned s0 = blockIdx.x, s1 = 3;
    u32x4 q = {1u, 2u, 3u, 4u};
    const unsigned la = wave * 16384 + lane * 16;
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = i;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
#define FILL(i) \
    if constexpr (KIND == 0) asm volatile("v_add_u32 %0, %0, 1" : "+v"(v[(i) & 7])); \
    if constexpr (KIND == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0)); \
    if constexpr (KIND == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(la), "n"(1024 * ((i) & 7))); \
    if constexpr (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[(i) & 7]) : "v"(f[7 - ((i) & 7)])); \
    if constexpr (KIND == 4) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[(i) & 7]) : "a"(c1[(i) & 7])); \
    if constexpr (KIND == 5) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(la), "v"(q), "n"(1024 * ((i) & 7)) : "memory"); \
    if constexpr (KIND == 6) { if (((i) & 1) == 0) asm volatile("v_add_u32 %0, %0, 1" : "+v"(v[(i) & 7])); else asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0)); } \
    if constexpr (KIND == 7) asm volatile("v_add_u32 %0, %0, %0" : "+v"(v[0]));  /* dependent chain */ \
    if constexpr (KIND == 8) asm volatile("s_nop 0");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < NF; ++i) { FILL(i) }
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < NF; ++i) { FILL(i + NF) }
        }
        if constexpr (KIND == 2 || KIND == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    for (int i = 0; i < 8; ++i) s += (float)v[i] + f[i];
    s += (float)(s0 + s1) + (float)(q[0] + q[1] + q[2] + q[3]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

int main() {
    const int nb = 256, iters = 400;
    float* out; long long* cyc;
    hipMalloc(&out, nb * 256 * sizeof(float));
    hipMalloc(&cyc, nb * 4 * sizeof(long long));
    std::vector<long long> h(nb * 4);
    auto run = [&](const char* name, auto kern, int nf) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
        hipMemcpy(h.data(), cyc, nb * 4 * sizeof(long long), hipMemcpyDeviceToHost);
        double c = 0; for (long long x : h) c += (double)x;
        printf("%-26s %2d per MFMA: %7.2f cycles per MFMA\n", name, nf, c / (nb * 4.0 * iters * 8));
    };
#define ROW(K, name) run(name, k<K, 0>, 0); run(name, k<K, 1>, 1); run(name, k<K, 2>, 2); run(name, k<K, 3>, 3); run(name, k<K, 4>, 4); run(name, k<K, 5>, 5); run(name, k<K, 6>, 6); run(name, k<K, 8>, 8); run(name, k<K, 12>, 12);
    ROW(0, "v_add_u32 (independent)")
    ROW(3, "v_fma_f32 (independent)")
    ROW(7, "v_add_u32 (dependent)")
    ROW(1, "s_add_u32")
    ROW(8, "s_nop 0")
    ROW(4, "v_accvgpr_read_b32")
    ROW(2, "ds_read_b128")
    ROW(5, "ds_write_b128")
    ROW(6, "v_add / s_add alternating")
    return 0;
}
