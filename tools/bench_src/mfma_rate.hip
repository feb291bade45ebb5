// Issue-rate microbenchmark of the f32 MFMA shapes on gfx950 (cycles per instruction per SIMD).
//   hipcc -O3 --offload-arch=gfx950 tools/bench_src/mfma_rate.hip -o gpurun_out/mfma_rate && gpurun_out/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

template <int SHAPE, int NACC>
__global__ void k(float* out, long long* cyc, int iters) {
    const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    long long t0 = 0, t1 = 0;
    if constexpr (SHAPE == 0) {          // 32x32x2
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        t1 = __builtin_amdgcn_s_memtime();
        float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else if constexpr (SHAPE == 1) {   // 16x16x4
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        t1 = __builtin_amdgcn_s_memtime();
        float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else if constexpr (SHAPE == 2) {   // 32x32x1 (2 blocks)
        f32x32 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 32; ++r) acc[i][r] = 0.f;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, acc[i], 0, 0, 0);
        t1 = __builtin_amdgcn_s_memtime();
        float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 32; ++r) s += acc[i][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else if constexpr (SHAPE == 3) {   // 16x16x1 (4 blocks)
        f32x16 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc[i], 0, 0, 0);
        t1 = __builtin_amdgcn_s_memtime();
        float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else if constexpr (SHAPE == 5) {   // 32x32x2, distinct A/B registers per accumulator (like a real kernel)
        f32x16 acc[NACC];
        float av[NACC], bv[NACC];
        for (int i = 0; i < NACC; ++i) { av[i] = a * (i + 1); bv[i] = b + i; for (int r = 0; r < 16; ++r) acc[i][r] = 0.f; }
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[i], av[i], acc[i], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime();
        float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = (t1 - t0) / 2;
        return;
    } else if constexpr (SHAPE == 6) {   // 16x16x4, distinct A/B registers
        f32x4 acc[NACC];
        float av[NACC], bv[NACC];
        for (int i = 0; i < NACC; ++i) { av[i] = a * (i + 1); bv[i] = b + i; for (int r = 0; r < 4; ++r) acc[i][r] = 0.f; }
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[i], av[i], acc[i], 0, 0, 0);
        }
        t1 = __builtin_amdgcn_s_memtime();
        float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = (t1 - t0) / 2;
        return;
    } else {                             // 4x4x1 (16 blocks)
        f32x4 acc[NACC];
        for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
        t1 = __builtin_amdgcn_s_memtime();
        float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int SHAPE, int NACC>
void run(const char* name, double macs, int threads, int blocks) {
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    const int nw = threads / 64 * blocks;
    hipMalloc(&cyc, sizeof(long long) * nw);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE, NACC><<<blocks, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<SHAPE, NACC><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nw); hipMemcpy(h.data(), cyc, sizeof(long long) * nw, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= nw;
    const double per_wave = mean / ((double)iters * NACC);                 // cycles per MFMA seen by one wave
    const int waves_per_simd = threads / 256;
    const double tf = 2.0 * macs * NACC * iters * nw / (ms * 1e-3) / 1e12;
    printf("%-10s NACC %d  waves/SIMD %d  blocks %4d: %.1f cycles/MFMA/wave -> %.1f cycles per MFMA per SIMD, %.1f TFLOP/s (events)\n",
           name, NACC, waves_per_simd, blocks, per_wave, per_wave / waves_per_simd, tf);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int threads : {256, 512}) {
        run<5, 4>("32x32x2 d", 2 * 2048, threads, 256);
        run<5, 8>("32x32x2 d", 2 * 2048, threads, 256);
        run<6, 16>("16x16x4 d", 2 * 1024, threads, 256);
        run<5, 4>("32x32x2 d", 2 * 2048, threads, 2048);
        run<0, 4>("32x32x2", 2048, threads, 2048);
        run<0, 4>("32x32x2", 2048, threads, 256);
        run<0, 8>("32x32x2", 2048, threads, 256);
        run<1, 4>("16x16x4", 1024, threads, 256);
        run<1, 16>("16x16x4", 1024, threads, 256);
        run<2, 4>("32x32x1", 2048, threads, 256);
        run<3, 8>("16x16x1", 1024, threads, 256);
        run<4, 16>("4x4x1", 256, threads, 256);
    }
    run<0, 4>("32x32x2", 2048, 256, 1);
    run<1, 8>("16x16x4", 1024, 256, 1);
    return 0;
}
