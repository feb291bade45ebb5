// Issue rate of v_pk_fma_f32 on gfx950 under the operand forms hipcc emits for the low-channel direct convolutions
// (cycles per wave-instruction on one SIMD).  hipcc -O3 --offload-arch=gfx950 tools/bench_src/pk_fma_rate.hip -o gpurun_out/pk_fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

// MODE 0: all-VGPR v_pk_fma_f32; 1: B = SGPR pair; 2: A broadcast (op_sel_hi 0) + SGPR pair; 3: plain v_fma_f32 (one FMA per lane);
// 4: A broadcast from hi half (op_sel:[1,0,0]) + SGPR pair
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters, const float* wsrc) {
    f2 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f2{0.f, 0.f};
    f2 a = f2{threadIdx.x * 0.001f, threadIdx.x * 0.002f};
    f2 bvec = f2{1.0f + threadIdx.x * 1e-4f, 1.0f};
    const float s0 = wsrc[0], s1 = wsrc[1];      // uniform -> SGPRs
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(bvec));
            if constexpr (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "s"(f2{s0, s1}));
            if constexpr (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(a), "s"(f2{s0, s1}));
            if constexpr (MODE == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc[i]) : "v"(a), "s"(f2{s0, s1}));
            if constexpr (MODE == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(a.x), "v"(bvec.x));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, int blocks) {
    float* out; long long* cyc; float* w;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipMalloc(&w, 64); float hw[2] = {1.0f, 0.5f}; hipMemcpy(w, hw, 8, hipMemcpyHostToDevice);
    const int nw = threads / 64 * blocks;
    hipMalloc(&cyc, sizeof(long long) * nw);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads>>>(out, cyc, iters, w);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, threads>>>(out, cyc, iters, w);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nw); hipMemcpy(h.data(), cyc, sizeof(long long) * nw, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= nw;
    const double per_wave = mean / ((double)iters * 8) ;          // s_memtime ticks (100 MHz) per instruction seen by one wave
    const double fl = (MODE == 3 ? 2.0 : 4.0) * 64 * 8.0 * iters * nw;
    printf("%-34s threads %4d blocks %5d: %.3f memtime ticks/instr/wave, %.1f TFLOP/s (events, %.3f ms)\n", name, threads, blocks, per_wave,
           fl / (ms * 1e-3) / 1e12, ms);
    hipFree(out); hipFree(cyc); hipFree(w);
}

int main() {
    for (int threads : {256, 512, 1024}) {
        run<0>("v_pk_fma_f32 vgpr,vgpr", threads, 2048);
        run<1>("v_pk_fma_f32 vgpr,sgpr", threads, 2048);
        run<2>("v_pk_fma_f32 bcast-lo,sgpr", threads, 2048);
        run<4>("v_pk_fma_f32 bcast-hi,sgpr", threads, 2048);
        run<3>("v_fma_f32", threads, 2048);
    }
    return 0;
}
