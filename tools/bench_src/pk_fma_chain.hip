// v_pk_fma_f32 under the conditions of the fused block1 kernel: 4 accumulator pairs (dependency distance 4), a different SGPR
// pair per instruction, broadcast A, 16 waves per CU (LDS-limited to two 512-thread workgroups).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int NACC, int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters, const float* wsrc) {
    extern __shared__ float lds[];
    f2 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f2{0.f, 0.f};
    f2 a[4];
    for (int i = 0; i < 4; ++i) a[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f};
    float w[32];
    for (int i = 0; i < 32; ++i) w[i] = wsrc[i];          // uniform -> SGPRs
    if (threadIdx.x == 9999) lds[0] = 1.f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = j % NACC;
            if constexpr (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(a[j & 3]), "s"(f2{w[2 * j], w[2 * j + 1]}));
            if constexpr (MODE == 1) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(a[j & 3].x), "s"(w[2 * j]));
                                       asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "v"(a[j & 3].x), "s"(w[2 * j + 1])); }
            if constexpr (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[j & 3]), "v"(a[(j + 1) & 3]));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC, int MODE>
void run(const char* name, int lds_bytes) {
    const int threads = 512, blocks = 2048;
    float* out; long long* cyc; float* w;
    (void)hipMalloc(&out, sizeof(float) * threads * blocks);
    (void)hipMalloc(&w, 256); std::vector<float> hw(64, 0.5f); (void)hipMemcpy(w, hw.data(), 256, hipMemcpyHostToDevice);
    const int nw = threads / 64 * blocks;
    (void)hipMalloc(&cyc, sizeof(long long) * nw);
    const int iters = 2000;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<NACC, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NACC, MODE><<<blocks, threads, lds_bytes>>>(out, cyc, iters, w);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NACC, MODE><<<blocks, threads, lds_bytes>>>(out, cyc, iters, w);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nw); (void)hipMemcpy(h.data(), cyc, sizeof(long long) * nw, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= nw;
    const double fl = 4.0 * 64 * 16.0 * iters * nw;
    printf("%-40s LDS %3d KB/WG: %.2f clk per pk-equivalent per wave, %.1f TFLOP/s (%s)\n", name, lds_bytes / 1024, mean / (iters * 16.0), fl / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
    (void)hipFree(out); (void)hipFree(cyc); (void)hipFree(w);
}

int main() {
    for (int lds : {0, 78 * 1024, 150 * 1024}) {
        run<4, 0>("pk_fma bcast,sgpr  4 acc pairs", lds);
        run<8, 0>("pk_fma bcast,sgpr  8 acc pairs", lds);
        run<4, 1>("2 x v_fma sgpr     4 acc pairs", lds);
        run<4, 2>("pk_fma vgpr,vgpr   4 acc pairs", lds);
    }
    return 0;
}
