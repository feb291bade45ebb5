// Do VALU instructions of one wave issue while another wave of the SAME SIMD keeps the matrix pipe busy?
//   hipcc -O3 --offload-arch=gfx950 tools/bench_src/coissue.hip -o gpurun_out/coissue && gpurun_out/coissue
// One workgroup of 8 waves per CU (waves w and w+4 share SIMD w&3).  Role of a wave: 0 = idle, 1 = MFMA stream
// (v_mfma_f32_32x32x2_f32, independent accumulators), 2 = VALU stream (dependent-free v_fma_f32), 3 = both interleaved
// in one wave (NV VALU ops after every MFMA).  Prints cycles per MFMA / per VALU op for each scenario.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters, int role_lo, int role_hi) {
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? role_lo : role_hi;
    const float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = a + i;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (role == 1) {
        for (int it = 0; it < iters; ++it) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
        }
    } else if (role == 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], b, a);
        }
    } else if (role == 3) {
        for (int it = 0; it < iters; ++it) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], b, a);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], b, a);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    const int nb = 256, iters = 20000;
    float* out; long long* cyc;
    hipMalloc(&out, nb * 512 * sizeof(float));
    hipMalloc(&cyc, nb * 8 * sizeof(long long));
    std::vector<long long> h(nb * 8);
    auto run = [&](const char* name, int lo, int hi, auto kern, int nv) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(kern, dim3(nb), dim3(512), 0, 0, out, cyc, iters, lo, hi);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), cyc, nb * 8 * sizeof(long long), hipMemcpyDeviceToHost);
        double lo_c = 0, hi_c = 0;
        for (int b = 0; b < nb; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? lo_c : hi_c) += (double)h[b * 8 + w];
        lo_c /= nb * 4.0 * iters; hi_c /= nb * 4.0 * iters;      // s_memtime ticks (100 MHz) per loop iteration
        printf("%-44s waves0-3: %8.4f ticks/iter   waves4-7: %8.4f ticks/iter  (NV=%d)\n", name, lo_c, hi_c, nv);
    };
    // one iteration = 2 MFMAs (role 1), 32 FMAs (role 2), 2 MFMAs + 2*NV FMAs (role 3)
    run("MFMA alone (1 wave/SIMD)", 1, 0, k<0>, 0);
    run("VALU alone (1 wave/SIMD)", 2, 0, k<0>, 0);
    run("MFMA wave + VALU wave on each SIMD", 1, 2, k<0>, 0);
    run("MFMA wave + MFMA wave on each SIMD", 1, 1, k<0>, 0);
    run("VALU wave + VALU wave on each SIMD", 2, 2, k<0>, 0);
    run("one wave: MFMA + 4 VALU interleaved", 3, 0, k<4>, 4);
    run("one wave: MFMA + 8 VALU interleaved", 3, 0, k<8>, 8);
    run("one wave: MFMA + 12 VALU interleaved", 3, 0, k<12>, 12);
    run("one wave: MFMA + 16 VALU interleaved", 3, 0, k<16>, 16);
    run("two waves: each MFMA + 8 VALU interleaved", 3, 3, k<8>, 8);
    return 0;
}
