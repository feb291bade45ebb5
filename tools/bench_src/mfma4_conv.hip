// Feasibility probe for block1's 8->8 3x3 stage on v_mfma_f32_4x4x1_16b_f32: every lane owns a pixel (B operand = its tap value from LDS,
// one ds_read_b32 with an immediate offset per tap), the weights sit in VGPRs (A operand = w[k][lane & 3], two cout quads), the 8 couts of the
// pixel accumulate in 2 x 4 registers.  Measures cycles per tap (ideal: two MFMAs = 16) at 1 / 2 waves per SIMD, and checks the layout
// against a scalar reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int C2H = 19, C2W = 35, C3H = 17, C3W = 33, PL = C2H * C2W;

template <int NIT>
__global__ __launch_bounds__(512) void k(const float* __restrict__ tile_g, const float* __restrict__ w /*[72][8]*/, float* __restrict__ out, long long* cyc, int reps) {
    __shared__ float T[8 * PL + 64];
    for (int e = threadIdx.x; e < 8 * PL; e += 512) T[e] = tile_g[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float wa[72], wb[72];
#pragma unroll
    for (int kk = 0; kk < 72; ++kk) { wa[kk] = w[kk * 8 + (lane & 3)]; wb[kk] = w[kk * 8 + 4 + (lane & 3)]; }
    long long t0 = __builtin_amdgcn_s_memtime();
    f32x4 lo[NIT], hi[NIT];
    for (int rep = 0; rep < reps; ++rep) {
        int base[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = min((wave * NIT + it) * 64 + lane, C3H * C3W - 1);
            const int r = e / C3W, c = e - r * C3W;
            base[it] = r * C2W + c;
            lo[it] = f32x4{0.f, 0.f, 0.f, 0.f}; hi[it] = lo[it];
        }
#pragma unroll
        for (int ci = 0; ci < 8; ++ci)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int kk = ci * 9 + t;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const float b = T[base[it] + ci * PL + (t / 3) * C2W + (t % 3)];
                    lo[it] = __builtin_amdgcn_mfma_f32_4x4x1f32(wa[kk], b, lo[it], 0, 0, 0);
                    hi[it] = __builtin_amdgcn_mfma_f32_4x4x1f32(wb[kk], b, hi[it], 0, 0, 0);
                }
            }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int e = (wave * NIT + it) * 64 + lane;
        if (e < C3H * C3W)
            for (int q = 0; q < 4; ++q) { out[((size_t)blockIdx.x * 8 + q) * 1024 + e] = lo[it][q]; out[((size_t)blockIdx.x * 8 + 4 + q) * 1024 + e] = hi[it][q]; }
    }
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    std::vector<float> ht(8 * PL), hw(72 * 8);
    for (size_t i = 0; i < ht.size(); ++i) ht[i] = sinf(0.37f * i) * 0.5f;
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = cosf(0.11f * i) * 0.3f;
    float *dt, *dw, *dout; long long* dc;
    const int blocks = 512;
    (void)hipMalloc(&dt, ht.size() * 4); (void)hipMalloc(&dw, hw.size() * 4); (void)hipMalloc(&dout, (size_t)blocks * 8 * 1024 * 4); (void)hipMalloc(&dc, blocks * 8 * 8);
    (void)hipMemcpy(dt, ht.data(), ht.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemset(dout, 0, (size_t)blocks * 8 * 1024 * 4);
    const int reps = 20;
    auto run = [&](auto kern, int nit, const char* name) {
        kern<<<blocks, 512>>>(dt, dw, dout, dc, reps); (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0); kern<<<blocks, 512>>>(dt, dw, dout, dc, reps); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> hc(blocks * 8); (void)hipMemcpy(hc.data(), dc, hc.size() * 8, hipMemcpyDeviceToHost);
        double m = 0; for (auto v : hc) m += v; m /= hc.size();
        const double taps = 72.0 * nit * reps;
        const double fl = 2.0 * 64 * 8 * taps * blocks * 8;
        printf("%s: %.1f cycles per tap per wave (2 MFMAs = 16 ideal; 2 waves/SIMD share the pipe -> 32), %.1f TFLOP/s (%s)\n", name, m / taps, fl / (ms * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
    };
    run(k<1>, 1, "1 position group per wave");
    run(k<2>, 2, "2 position groups per wave (4 accumulator chains)");
    // layout check of the last run (NIT = 2 covers positions 0..1023 > 561)
    std::vector<float> ho(8 * 1024); (void)hipMemcpy(ho.data(), dout, ho.size() * 4, hipMemcpyDeviceToHost);
    double maxd = 0;
    for (int e = 0; e < C3H * C3W; ++e) {
        const int r = e / C3W, c = e % C3W;
        for (int co = 0; co < 8; ++co) {
            float acc = 0.f;
            for (int ci = 0; ci < 8; ++ci) for (int t = 0; t < 9; ++t) acc = fmaf(ht[ci * PL + (r + t / 3) * C2W + c + t % 3], hw[(ci * 9 + t) * 8 + co], acc);
            maxd = fmax(maxd, fabs(acc - ho[co * 1024 + e]));
        }
    }
    printf("max |mfma - fmaf chain| over %d positions x 8 couts: %g (0 = bit-identical)\n", C3H * C3W, maxd);
    return 0;
}
