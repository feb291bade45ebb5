// The same stream as l2_stream.hip through the LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave-instruction, 8 in flight per wave): B/clk/CU.
//   hipcc -O2 --offload-arch=gfx950 tools/bench_src/l2_stream_dma.hip -o gpurun_probe/l2_stream_dma ; gpurun_probe/l2_stream_dma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(NT) void k(const unsigned char* __restrict__ src, unsigned span, int iters, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long ba = (unsigned long long)src;
    i4 rs;
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ba);
    rs.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(ba >> 32) & 0xffffu));
    rs.z = __builtin_amdgcn_readfirstlane((int)span);
    rs.w = 0x00020000;
    const unsigned l0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds + wave * 8192;
    unsigned pos = (unsigned)(((size_t)blockIdx.x * 4099 * NT * 16 + wave * 8192) % span);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned p = pos + j * 1024;
            if (p >= span) p -= span;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(__builtin_amdgcn_readfirstlane((int)(l0 + j * 1024))), "v"(lane * 16), "s"(rs), "s"(__builtin_amdgcn_readfirstlane((int)p)) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        pos += NT / 64 * 8192;
        if (pos >= span) pos -= span;
    }
    __syncthreads();
    if (lds[threadIdx.x] == 0x77 && iters < 0) out[0] = 1;
}

int main() {
    const size_t bytes = 512u << 20;
    unsigned char* src; unsigned* out;
    hipMalloc(&src, bytes); hipMemset(src, 1, bytes); hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int nt, int wgs, size_t span, int iters) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(nt), nt / 64 * 8192, 0, src, (unsigned)span, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double b = (double)wgs * (nt / 64) * 8192.0 * iters;
        printf("%-22s %4d threads x %4d workgroups, span %8.2f MB: %8.1f GB/s  = %6.1f B/clk/CU at 2.4 GHz\n", name, nt, wgs, span / 1048576.0, b / ms / 1e6, b / ms / 1e6 / 256 / 2.4);
    };
    for (size_t span : {(size_t)48 << 10, (size_t)1536 << 10, (size_t)16 << 20, (size_t)400 << 20}) {
        run("LDS-DMA, 4 waves/CU", k<256>, 256, 256, span, 2000);
        run("LDS-DMA, 8 waves/CU", k<512>, 512, 256, span, 1000);
    }
    return 0;
}
