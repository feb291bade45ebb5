// Does v_mfma_f32_32x32x16_f16 issue at its 32-cycle rate whatever register file its A / B operands come from?  (linear_fx_body.hpp keeps every operand in accumulation
// registers.)  Four accumulators in turn, everything in inline asm; one or two waves per SIMD.  Prints shader cycles (s_memtime) per MFMA and per SIMD.
//   hipcc -O2 --offload-arch=gfx950 tools/bench_src/mfma_operands.hip -o gpurun_probe/mfma_operands ; gpurun_probe/mfma_operands
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 c[4];
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) c[j][r] = 0.f;
    f16x8 a[2], b[2];
    for (int j = 0; j < 2; ++j)
        for (int r = 0; r < 8; ++r) { a[j][r] = (_Float16)(0.001f * lane + j); b[j][r] = (_Float16)(0.002f * r + j); }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
#define MF(ci, ai, bi)                                                                                                                      \
    if constexpr (KIND == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[ci]) : "v"(a[ai]), "v"(b[bi]));             \
    if constexpr (KIND == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[ci]) : "a"(a[ai]), "v"(b[bi]));             \
    if constexpr (KIND == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[ci]) : "v"(a[ai]), "a"(b[bi]));             \
    if constexpr (KIND == 3) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[ci]) : "a"(a[ai]), "a"(b[bi]));             \
    if constexpr (KIND == 4) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[ci]) : "v"(a[ai]), "v"(b[bi]));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) { MF(0, 0, 0) MF(1, 1, 0) MF(2, 0, 1) MF(3, 1, 1) }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += c[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

int main() {
    const int iters = 400;
    float* out; long long* cyc;
    hipMalloc(&out, 512 * 256 * sizeof(float));
    hipMalloc(&cyc, 512 * 4 * sizeof(long long));
    std::vector<long long> h(512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int nb) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, 0, out, cyc, iters * (rep ? 50 : 1)); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); }
        hipMemcpy(h.data(), cyc, nb * 4 * sizeof(long long), hipMemcpyDeviceToHost);
        double c = 0; for (int i = 0; i < nb * 4; ++i) c += (double)h[i];
        const double per = c / (nb * 4.0 * iters * 50 * 8);
        // (the second launch runs 50 x as long: its s_memtime count against the wall clock of the launch = the rate the counter ticks at, with every matrix pipe of the chip busy)
        printf("%-34s %d wave(s) per SIMD: %7.2f counts per MFMA of a wave, %7.2f per MFMA of the SIMD; %.0f counts in %.3f ms = %.2f GHz\n", name, nb / 256, per, per / (nb / 256),
               c / (nb * 4.0), ms, c / (nb * 4.0) / ms / 1e6);
    };
    for (int nb : {256, 512}) {
        run("A, B in VGPRs, C in AGPRs", k<0>, nb);
        run("A in AGPRs, B in VGPRs", k<1>, nb);
        run("A in VGPRs, B in AGPRs", k<2>, nb);
        run("A, B, C in AGPRs", k<3>, nb);
        run("A, B, C in VGPRs", k<4>, nb);
    }
    return 0;
}
