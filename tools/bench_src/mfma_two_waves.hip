// What ONE multiplying wave per SIMD leaves on the table, and what a second one picks up: v_mfma_f32_32x32x16_f16 streams of conv_rs64_kernel's shape (per 32-position block
// and SIMD: 54 MFMAs in 9 taps, each tap's two B fragments read from LDS two taps ahead, two accumulator chains that take turns) issued
//   A0 / A1 : by one wave per SIMD (all 54 MFMAs; A0 without the reads),
//   B0 / B1 : by two waves per SIMD that split the COUT blocks (27 MFMAs each; every wave reads every tap's fragments: LDS reads x 2),
//   B2      : by two waves per SIMD that split the TAPS (5 + 4 taps of 6 MFMAs; no read is issued twice),
//   C1      : like A1 with six more ds_read_b128 and six ds_write_b128 per block (the four-way reduction of the K split), D1 / D2: the same beside B1 / B2 (3 + 3 per wave).
// One workgroup per CU (LDS), 256 CUs; the time per block and SIMD against A0's = what the stream keeps of the matrix pipe.  Everything in inline asm (program order = issue order).
//   hipcc -O2 --offload-arch=gfx950 tools/bench_src/mfma_two_waves.hip -o gpurun_probe/mfma_two_waves && gpurun_probe/mfma_two_waves
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MF(c, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(A), "v"(B))
#define RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define WR(addr, src, off) asm volatile("ds_write_b128 %0, %1 offset:" #off :: "v"(addr), "v"(src) : "memory")
#define LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")

// MPT: MFMAs per tap (6: both cout blocks, 3: one); NT: taps of this wave per block; READS: issue the two reads per tap; RED: reduction traffic (reads + writes per block)
template <int MPT, int NT, bool READS, int RED>
__device__ __forceinline__ void block_stream(f32x16& c0, f32x16& c1, const f16x8 (&w)[3], f16x8 (&x)[3][2], unsigned la, u32x4& q) {
    u32x4 q2[2] = {q, q};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int cur = t % 3, far = (t + 2) % 3;
        if constexpr (READS) { RD(x[far][0], la, 0); RD(x[far][1], la, 1024); LGKM(4); }      // tap t + 2 requested, tap t's fragments (two taps old) awaited
        MF(c0, w[0], x[cur][0]);
        if constexpr (MPT == 6) MF(c1, w[0], x[cur][0]);
        MF(MPT == 6 ? c0 : c1, w[1], x[cur][1]);
        if constexpr (MPT == 6) MF(c1, w[1], x[cur][1]);
        MF(c0, w[2], x[cur][0]);
        if constexpr (MPT == 6) MF(c1, w[2], x[cur][0]);
        if constexpr (RED > 0 && RED < 100) { if (t == 2) { for (int i = 0; i < RED; ++i) WR(la, q, 2048); } if (t == 4) { for (int i = 0; i < RED; ++i) RD(q, la, 2048); LGKM(0); } }
        // RED = 100 + n: the same n stores and n reads, the reads awaited two taps later (behind 12 more MFMAs; LDS returns in order: the four operand reads issued since stay out);
        // RED = 200 + n: only the n stores; RED = 300 + n: only the n reads (awaited two taps later)
        if constexpr (RED >= 100) { constexpr int n = RED % 100, kind = RED / 100;
            if (t == 2 && kind != 3) { for (int i = 0; i < n; ++i) WR(la, q, 2048); }
            if (t == 4 && kind != 2) { for (int i = 0; i < n; ++i) RD(q2[i & 1], la, 2048); }
            if (t == 6 && kind != 2) { LGKM(4); q[0] += q2[0][0] + q2[1][0]; } }
    }
}

// conv_bx64s2x_kernel's shapes: the WEIGHT fragments come from LDS too.  NR reads (X and W fragments) + NM MFMAs per K step, the reads one step ahead
template <int NR, int NM>
__device__ __forceinline__ void step_stream(f32x16& c0, f32x16& c1, f16x8 (&x)[3][2], f16x8 (&y)[2][10], unsigned la) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < NR; ++i) { if (i == 0) RD(y[s ^ 1][0], la, 0); else if (i == 1) RD(y[s ^ 1][1], la, 1024); else if (i == 2) RD(y[s ^ 1][2], la, 2048); else if (i == 3) RD(y[s ^ 1][3], la, 3072);
            else if (i == 4) RD(y[s ^ 1][4], la, 4096); else if (i == 5) RD(y[s ^ 1][5], la, 5120); else if (i == 6) RD(y[s ^ 1][6], la, 6144); else if (i == 7) RD(y[s ^ 1][7], la, 7168);
            else if (i == 8) RD(y[s ^ 1][8], la, 0); else RD(y[s ^ 1][9], la, 1024); }
        if constexpr (NR == 5) LGKM(5); else if constexpr (NR == 8) LGKM(8); else if constexpr (NR == 7) LGKM(7); else if constexpr (NR == 4) LGKM(4); else LGKM(10);
#pragma unroll
        for (int m = 0; m < NM; ++m) { if ((m + s * NM) & 1) MF(c1, y[s][m % NR], y[s][(m + 1) % NR]); else MF(c0, y[s][m % NR], y[s][(m + 1) % NR]); }      // (the two chains take turns across the steps too)
    }
}

template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 c0, c1;
    for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
    f16x8 w[3], x[3][2];
    for (int j = 0; j < 3; ++j) for (int r = 0; r < 8; ++r) { w[j][r] = (_Float16)(0.001f * (lane + j)); x[j][0][r] = (_Float16)(0.002f * r); x[j][1][r] = (_Float16)(0.003f * r); }
    u32x4 q = {1u, 2u, 3u, 4u};
    f16x8 y[2][10];
    for (int a = 0; a < 2; ++a) for (int j = 0; j < 10; ++j) for (int r = 0; r < 8; ++r) y[a][j][r] = (_Float16)(0.001f * (r + j));
    const unsigned la = wave * 8192 + lane * 16;
    for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) block_stream<6, 9, false, 0>(c0, c1, w, x, la, q);            // A0
        if constexpr (KIND == 1) block_stream<6, 9, true, 0>(c0, c1, w, x, la, q);             // A1
        if constexpr (KIND == 2) block_stream<3, 9, false, 0>(c0, c1, w, x, la, q);            // B0
        if constexpr (KIND == 3) block_stream<3, 9, true, 0>(c0, c1, w, x, la, q);             // B1
        if constexpr (KIND == 4) { if (wave & 4) block_stream<6, 4, true, 0>(c0, c1, w, x, la, q); else block_stream<6, 5, true, 0>(c0, c1, w, x, la, q); }      // B2 (waves w, w + 4 share a SIMD)
        if constexpr (KIND == 5) block_stream<6, 9, true, 6>(c0, c1, w, x, la, q);             // C1
        if constexpr (KIND == 6) block_stream<3, 9, true, 3>(c0, c1, w, x, la, q);             // D1
        // (54 MFMAs per SIMD and "block" in every variant: 4.5 double steps of 3 MFMAs on each of two waves, 4.5 of 6 or 2.25 of 12 on one)
        if constexpr (KIND == 8) { for (int z = 0; z < 4; ++z) step_stream<5, 3>(c0, c1, x, y, la); if (it & 1) step_stream<5, 3>(c0, c1, x, y, la); }            // S1: two waves per SIMD, 5 reads + 3 MFMAs per step (the eight-wave form)
        if constexpr (KIND == 9) { for (int z = 0; z < 4; ++z) step_stream<8, 6>(c0, c1, x, y, la); if (it & 1) step_stream<8, 6>(c0, c1, x, y, la); }      // S2: one wave per SIMD, 8 reads + 6 MFMAs (4.5 double steps)
        if constexpr (KIND == 10) { for (int z = 0; z < 4; ++z) step_stream<7, 6>(c0, c1, x, y, la); if (it & 1) step_stream<7, 6>(c0, c1, x, y, la); }     // S3: one wave, two pixel blocks x one cout block: 7 reads + 6 MFMAs
        if constexpr (KIND == 11) { for (int z = 0; z < 2; ++z) step_stream<10, 12>(c0, c1, x, y, la); if ((it & 3) == 0) step_stream<10, 12>(c0, c1, x, y, la); }      // S4: one wave, 2 x 2 tile: 10 reads + 12 MFMAs (2.25 double steps)
        if constexpr (KIND == 12) { for (int z = 0; z < 4; ++z) step_stream<4, 3>(c0, c1, x, y, la); if (it & 1) step_stream<4, 3>(c0, c1, x, y, la); }           // S5: two waves per SIMD, 4 reads + 3 MFMAs (fp16(w) derived in registers)
        if constexpr (KIND == 13) block_stream<6, 9, true, 106>(c0, c1, w, x, la, q);          // C2
        if constexpr (KIND == 14) block_stream<6, 9, true, 206>(c0, c1, w, x, la, q);          // C3
        if constexpr (KIND == 15) block_stream<6, 9, true, 306>(c0, c1, w, x, la, q);          // C4
        if constexpr (KIND == 7) { if (wave & 4) block_stream<6, 4, true, 3>(c0, c1, w, x, la, q); else block_stream<6, 5, true, 3>(c0, c1, w, x, la, q); }      // D2
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    float s = (float)q[0];
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    for (int j = 0; j < 3; ++j) s += (float)x[j][0][0] + (float)x[j][1][0];
    for (int j = 0; j < 10; ++j) s += (float)y[0][j][0] + (float)y[1][j][0];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static double run(const char* name, int threads, float* out, int iters, double base) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<256, threads, 100 * 1024>>>(out, iters / 10);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0); k<KIND><<<256, threads, 100 * 1024>>>(out, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double ns_per_block = 1e6 * best / iters;      // one block = 54 MFMAs per SIMD in every variant
    printf("%-78s %8.1f ns per block and SIMD  (%5.1f ns per MFMA)   pipe kept vs A0: %5.1f %%\n", name, ns_per_block, ns_per_block / 54, base > 0 ? 100.0 * base / ns_per_block : 100.0);
    return ns_per_block;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    const double a0 = run<0>("A0  one wave per SIMD, 54 MFMAs, no reads", 256, out, iters, 0);
    run<1>("A1  one wave per SIMD, 54 MFMAs + 18 ds_read_b128 (conv_rs64_kernel's stream)", 256, out, iters, a0);
    run<5>("C1  A1 + 6 reads + 6 writes per block (the K split's reduction)", 256, out, iters, a0);
    run<13>("C2  C1 with the six reads awaited two taps later", 256, out, iters, a0);
    run<14>("C3  A1 + the six ds_write_b128 only", 256, out, iters, a0);
    run<15>("C4  A1 + the six reads only (awaited two taps later)", 256, out, iters, a0);
    run<2>("B0  two waves per SIMD, 27 MFMAs each, no reads", 512, out, iters, a0);
    run<3>("B1  two waves per SIMD split the cout blocks: 27 MFMAs + 18 reads each", 512, out, iters, a0);
    run<6>("D1  B1 + 3 reads + 3 writes per wave and block", 512, out, iters, a0);
    run<4>("B2  two waves per SIMD split the taps: 30 + 24 MFMAs, 10 + 8 reads", 512, out, iters, a0);
    run<7>("D2  B2 + 3 reads + 3 writes per wave and block", 512, out, iters, a0);
    printf("conv_bx64s2x_kernel's shapes (weights from LDS as well):\n");
    run<8>("S1  two waves per SIMD, 5 reads + 3 MFMAs per K step (1.67 reads per MFMA: the eight multiplying waves)", 512, out, iters, a0);
    run<12>("S5  two waves per SIMD, 4 reads + 3 MFMAs (fp16(w) derived: 1.33)", 512, out, iters, a0);
    run<9>("S2  one wave per SIMD, 8 reads + 6 MFMAs (one pixel block x two cout blocks: 1.33)", 256, out, iters, a0);
    run<10>("S3  one wave per SIMD, 7 reads + 6 MFMAs (two pixel blocks x one cout block: 1.17)", 256, out, iters, a0);
    run<11>("S4  one wave per SIMD, 10 reads + 12 MFMAs (2 x 2 tile: 0.83)", 256, out, iters, a0);
    return 0;
}
