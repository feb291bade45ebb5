// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs (and what does v_cvt_pk_f16_f32 do below 2^-14)?
//   hipcc -O3 --offload-arch=gfx950 tools/bench_src/f16_denorm.hip -o /tmp/f16_denorm && /tmp/f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float a_val, float b_val, float* out) {
    // A[i][k] = a_val for k == 0 (else 0), B[k][j] = b_val for k == 0: D[i][j] = a_val * b_val
    const int lane = threadIdx.x;
    f2 av = {a_val, 0.f}, bv = {b_val, 0.f};
    h2 ah = __builtin_convertvector(av, h2), bh = __builtin_convertvector(bv, h2);
    h8 a = {}, b = {};
    if (lane < 32) { a[0] = ah[0]; b[0] = bh[0]; }      // k = 0 lives in lanes 0..31, element 0
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) { out[0] = acc[0]; out[1] = (float)ah[0]; out[2] = (float)bh[0]; }
}
int main() {
    float* d; hipMalloc(&d, 16);
    const float cases[][2] = {{1.f, 1.f}, {3.0e-5f, 1024.f}, {1024.f, 3.0e-5f}, {6.0e-8f, 16384.f}, {16384.f, 6.0e-8f}, {3.0e-5f, 3.0e-5f}, {6.2e-5f, 1.f}, {1e-7f, 60000.f}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, 1, 64, 0, 0, c[0], c[1], d);
        float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a %.6g (as f16 %.9g)  b %.6g (as f16 %.9g): mfma a*b = %.9g   exact product of the f16 values %.9g\n", c[0], h[1], c[1], h[2], h[0], (double)h[1] * h[2]);
    }
    return 0;
}
