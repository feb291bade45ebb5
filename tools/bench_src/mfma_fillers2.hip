// Round 2 of tools/bench_src/mfma_fillers.hip: a lone wave per SIMD, v_mfma_f32_32x32x16_f16 stream in groups of SIX (a tap of conv_rs64_kernel), memory instructions at the
// density the kernel has them.  Cycles per group of six MFMAs (floor 192).  Everything in inline asm (program order = issue order).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define MF(c, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(A), "v"(B))
#define MFA(c, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "a"(A), "a"(B))

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float* gbuf, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 c0, c1, d0;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; d0[r] = 1.f + r; }
    f16x8 a, b, xa, xb, ya, yb;
    for (int r = 0; r < 8; ++r) { a[r] = (_Float16)(0.001f * lane); b[r] = (_Float16)(0.002f * r); xa[r] = a[r]; xb[r] = b[r]; ya[r] = a[r]; yb[r] = b[r]; }
    asm volatile("" : "+a"(xa), "+a"(xb), "+a"(ya), "+a"(yb), "+a"(d0));
    unsigned v[8];
    for (int i = 0; i < 8; ++i) v[i] = lane + i;
    u32x4 q = {1u, 2u, 3u, 4u}, q2 = q;
    u32x2 h2 = {5u, 6u};
    float g0 = 0.f, g1 = 0.f;
    const unsigned la = wave * 16384 + lane * 16, la8 = wave * 16384 + lane * 8;
    float* gp = gbuf + ((size_t)blockIdx.x * 256 + threadIdx.x);
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<unsigned*>(lds)[i] = i;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // ---- one group: six MFMAs, the KIND's extras
        if constexpr (KIND == 1 || KIND == 2) { asm volatile("ds_read_b128 %0, %1" : "=a"(xa) : "v"(la)); asm volatile("ds_read_b128 %0, %1 offset:1024" : "=a"(xb) : "v"(la)); }      // the next tap's operands, up front
        if constexpr (KIND == 9) { asm volatile("ds_read_b128 %0, %1" : "=v"(q) : "v"(la)); asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(q2) : "v"(la)); }
        if constexpr (KIND == 2 || KIND == 10) { MFA(c0, ya, yb); MFA(c1, ya, yb); } else { MF(c0, a, b); MF(c1, a, b); }
        if constexpr (KIND == 3) asm volatile("ds_write_b128 %0, %1" :: "v"(la), "v"(q) : "memory");
        if constexpr (KIND == 4) { asm volatile("ds_write_b64 %0, %1" :: "v"(la8), "v"(h2) : "memory"); asm volatile("ds_write_b64 %0, %1 offset:8192" :: "v"(la8), "v"(h2) : "memory"); }
        if constexpr (KIND == 5) asm volatile("buffer_load_dword %0, %1, s[0:3], 0 offen" : "=v"(g0) : "v"(0x80000000u));   /* placeholder, replaced below */
        if constexpr (KIND == 6) asm volatile("global_load_dword %0, %1, off" : "=v"(g0) : "v"(gp));
        if constexpr (KIND == 7) asm volatile("global_store_dword %0, %1, off" :: "v"(gp), "v"(g1) : "memory");
        if constexpr (KIND == 8) { asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[0]) : "a"(d0[0])); asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[1]) : "a"(d0[1])); }
        if constexpr (KIND == 11) asm volatile("ds_write_b128 %0, %1" :: "v"(la), "a"(d0[0]) : "memory");   /* placeholder */
        if constexpr (KIND == 2 || KIND == 10) { MFA(c0, ya, yb); MFA(c1, ya, yb); } else { MF(c0, a, b); MF(c1, a, b); }
        if constexpr (KIND == 12) { asm volatile("ds_write_b128 %0, %1" :: "v"(la), "v"(q) : "memory"); }
        if constexpr (KIND == 6) asm volatile("global_load_dword %0, %1, off offset:1024" : "=v"(g1) : "v"(gp));
        if constexpr (KIND == 7) asm volatile("global_store_dword %0, %1, off offset:1024" :: "v"(gp), "v"(g1) : "memory");
        if constexpr (KIND == 8) { asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[2]) : "a"(d0[2])); asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[3]) : "a"(d0[3])); }
        if constexpr (KIND == 2 || KIND == 10) { MFA(c0, ya, yb); MFA(c1, ya, yb); } else { MF(c0, a, b); MF(c1, a, b); }
        if constexpr (KIND == 12) { asm volatile("ds_write_b128 %0, %1 offset:1024" :: "v"(la), "v"(q) : "memory"); }
        if constexpr (KIND == 1 || KIND == 2 || KIND == 9) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (KIND == 3 || KIND == 4 || KIND == 12) { if ((it & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        if constexpr (KIND == 6 || KIND == 7) { if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float s = g0 + g1;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + d0[r];
    for (int i = 0; i < 8; ++i) s += (float)v[i] + (float)xa[i] + (float)xb[i];
    s += (float)(q[0] + q2[1] + h2[0]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

int main() {
    const int nb = 256, iters = 600;
    float *out, *gbuf; long long* cyc;
    hipMalloc(&out, nb * 256 * sizeof(float));
    hipMalloc(&gbuf, (size_t)nb * 256 * 4 + 8192);
    hipMemset(gbuf, 0, (size_t)nb * 256 * 4 + 8192);
    hipMalloc(&cyc, nb * 4 * sizeof(long long));
    std::vector<long long> h(nb * 4);
    auto run = [&](const char* name, auto kern) {
        for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(kern, dim3(nb), dim3(256), 0, 0, out, gbuf, cyc, iters); hipDeviceSynchronize(); }
        hipMemcpy(h.data(), cyc, nb * 4 * sizeof(long long), hipMemcpyDeviceToHost);
        double c = 0; for (long long x : h) c += (double)x;
        printf("%-86s %7.1f cycles per six MFMAs\n", name, c / (nb * 4.0 * iters));
    };
    run("six MFMAs alone (operands in VGPRs)", k<0>);
    run("six MFMAs alone, A and B operands in AGPRs", k<10>);
    run("+ 2 ds_read_b128 into AGPRs up front, awaited behind the six (VGPR-operand MFMAs)", k<1>);
    run("+ 2 ds_read_b128 into AGPRs up front, awaited behind the six (AGPR-operand MFMAs)", k<2>);
    run("+ 2 ds_read_b128 into VGPRs up front, awaited behind the six", k<9>);
    run("+ 1 ds_write_b128 behind the first pair", k<3>);
    run("+ 2 ds_write_b128, one behind the second and one behind the third pair", k<12>);
    run("+ 2 ds_write_b64 behind the first pair", k<4>);
    run("+ 2 global_load_dword (one behind the first, one behind the second pair)", k<6>);
    run("+ 2 global_store_dword (likewise)", k<7>);
    run("+ 4 v_accvgpr_read_b32 of an idle accumulator (two behind the first, two behind the second pair)", k<8>);
    return 0;
}
