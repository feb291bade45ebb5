// What paces the K loop of conv_bx_kernel: 12 bf16 MFMAs (32x32x16, two accumulator chains) and NRD ds_read_b128 per step.
//   hipcc -O3 --offload-arch=gfx950 tools/bench_src/bx_loop_rate.hip -o gpurun_out/bx_loop_rate && gpurun_out/bx_loop_rate
// Reports cycles per step (ideal 12 x 32 = 384 per wave, x waves per SIMD) for: chains per wave 1 / 2 / 4, reads per step 0 / 4 / 8,
// 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NCH, int NRD>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5, wave = tid >> 6;
    for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0x3f803f80u + i;      // 64 KiB of finite bf16 pairs
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 w[3], x[8];
    for (int q = 0; q < 3; ++q) w[q] = *reinterpret_cast<const bf16x8*>(lds + lane * 16 + q * 1024);
    for (int q = 0; q < 8; ++q) x[q] = *reinterpret_cast<const bf16x8*>(lds + lane * 16 + q * 1024 + 4096);
    const unsigned char* p = lds + ((wave & 3) * 2 * 34 + l31) * 144 + half * 16;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    bf16x8 y[8];
    for (int q = 0; q < 8; ++q) y[q] = x[q];
    for (int it = 0; it < iters; it += 2) {      // ping-pong x / y: no register moves
#pragma unroll
        for (int q = 0; q < NRD; ++q) y[q] = *reinterpret_cast<const bf16x8*>(p + (it & 6) * 432 + (q >> 2) * 4896 + (q & 3) * 48);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 12; ++m) acc[m % NCH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[m % 3], x[(m * 5) % 8], acc[m % NCH], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NRD; ++q) x[q] = *reinterpret_cast<const bf16x8*>(p + ((it + 1) & 7) * 432 + (q >> 2) * 4896 + (q & 3) * 48);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 12; ++m) acc[m % NCH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[m % 3], y[(m * 5) % 8], acc[m % NCH], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + tid] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}

template <int NCH, int NRD>
static void run(int threads, float* out, long long* cyc, int iters) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<NCH, NRD>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    const int grid = 256;
    k<NCH, NRD><<<grid, threads, 70 * 1024>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    k<NCH, NRD><<<grid, threads, 70 * 1024>>>(out, cyc, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(grid * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double m = 0; for (auto v : h) m += v; m /= h.size();
    printf("chains %d reads/step %d waves/SIMD %d: %.1f cycles per step per wave (s_memtime), %.1f us\n", NCH, NRD, threads / 256, m / iters, ms * 1e3);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 2000;
    for (int threads : {256, 512}) {
        run<1, 0>(threads, out, cyc, iters); run<2, 0>(threads, out, cyc, iters); run<4, 0>(threads, out, cyc, iters);
        run<2, 4>(threads, out, cyc, iters); run<2, 8>(threads, out, cyc, iters); run<4, 8>(threads, out, cyc, iters);
    }
    return 0;
}
