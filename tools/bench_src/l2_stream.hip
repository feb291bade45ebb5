// How many bytes per clock does a CU get out of its XCD's L2 when every CU of the chip streams L2-RESIDENT data (linear_fx_body.hpp: the fragments of a layer, 1.5 MB, and the
// rows its column blocks share)?  Every workgroup (4 or 8 waves) reads `span` bytes round and round, 8 x 16 B in flight per lane; prints GB/s of the chip and B/clk/CU.
//   hipcc -O2 --offload-arch=gfx950 tools/bench_src/l2_stream.hip -o gpurun_probe/l2_stream ; gpurun_probe/l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(NT) void k(const u4* __restrict__ src, size_t span_u4, int iters, unsigned* out) {
    u4 acc = {0u, 0u, 0u, 0u};
    size_t pos = ((size_t)blockIdx.x * 4099 * NT) % span_u4;      // (workgroups start at different places)
    for (int it = 0; it < iters; ++it) {
        u4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            size_t p = pos + (size_t)j * NT + threadIdx.x;
            if (p >= span_u4) p -= span_u4;
            v[j] = __builtin_nontemporal_load(src + p) ;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j];
        pos += 8 * NT;
        if (pos >= span_u4) pos -= span_u4;
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}
template <int NT>
__global__ __launch_bounds__(NT) void kp(const u4* __restrict__ src, size_t span_u4, int iters, unsigned* out) {      // (plain loads)
    u4 acc = {0u, 0u, 0u, 0u};
    size_t pos = ((size_t)blockIdx.x * 4099 * NT) % span_u4;
    for (int it = 0; it < iters; ++it) {
        u4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            size_t p = pos + (size_t)j * NT + threadIdx.x;
            if (p >= span_u4) p -= span_u4;
            v[j] = src[p];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j];
        pos += 8 * NT;
        if (pos >= span_u4) pos -= span_u4;
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

int main() {
    const size_t bytes = 512u << 20;
    u4* src; unsigned* out;
    hipMalloc(&src, bytes); hipMemset(src, 1, bytes); hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int nt, int wgs, size_t span, int iters) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kern, dim3(wgs), dim3(nt), 0, 0, src, span / 16, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double b = (double)wgs * nt * 128.0 * iters;
        printf("%-22s %4d threads x %4d workgroups, span %8.2f MB: %8.1f GB/s  = %6.1f B/clk/CU at 2.4 GHz\n", name, nt, wgs, span / 1048576.0, b / ms / 1e6, b / ms / 1e6 / 256 / 2.4);
    };
    for (size_t span : {(size_t)48 << 10, (size_t)1536 << 10, (size_t)16 << 20, (size_t)400 << 20}) {
        run("plain, 4 waves/CU", kp<256>, 256, 256, span, 2000);
        run("plain, 8 waves/CU", kp<512>, 512, 256, span, 1000);
        run("plain, 16 waves/CU", kp<512>, 512, 512, span, 1000);
        run("nontemporal, 8 waves/CU", k<512>, 512, 256, span, 1000);
    }
    return 0;
}
