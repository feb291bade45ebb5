// Stand-alone reproducer attempt for the round-4 finding: a group of v_mfma_f32_32x32x16_bf16 whose B operands are vector-ALU results, executed by two
// waves per SIMD, the FIRST time through its code (instruction-cache refills), delivers wrong results for columns 16..31.
//
// One K = 64 layer of the split-bf16 key-point head, reduced to what the failure needs: per wave 32 "cells" x 64 inputs from global memory, three-way bf16 split in
// registers, 4 K steps x 12 MFMAs on two accumulators with the weight fragments in LDS -- the code of head_bx_layer<2> (k_heads.hip) -- executed TWICE by every wave
// (a rolled loop: the same instruction addresses): pass 0 on the cold cache (s_icache_inv at kernel start), pass 1 warm.  The kernel compares the two accumulator
// sets itself; a difference is a wrong pass 0.  SHIFT moves the body by 4 x SHIFT bytes against the 64-byte instruction lines.
//     hipcc -O3 --offload-arch=gfx950 tools/bench_src/mfma_refill.hip -o /tmp/mfma_refill && /tmp/mfma_refill [seconds per position]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int N> __device__ inline void code_shift() {
    if constexpr (N > 0) { asm volatile("s_nop 0"); code_shift<N - 1>(); }
}
__device__ inline unsigned pk_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ inline void split8(const float (&y)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    uint4 uh, um, ul;
    unsigned* ph = &uh.x; unsigned* pm = &um.x; unsigned* pl = &ul.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = y[2 * i], b = y[2 * i + 1];
        const unsigned hh = pk_bf16(a, b);
        const float ra = a - __uint_as_float(hh << 16), rb = b - __uint_as_float(hh & 0xffff0000u);
        const unsigned mm = pk_bf16(ra, rb);
        const float sa = ra - __uint_as_float(mm << 16), sb = rb - __uint_as_float(mm & 0xffff0000u);
        ph[i] = hh; pm[i] = mm; pl[i] = pk_bf16(sa, sb);
    }
    h = __builtin_bit_cast(bf16x8, uh); m = __builtin_bit_cast(bf16x8, um); l = __builtin_bit_cast(bf16x8, ul);
}

constexpr int L_BYTES = 2 * 4 * 3 * 1024;      // [K step 4][cout block 2][split 3][64 lanes] 16 B

template <int SHIFT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void refill_kernel(const float* __restrict__ x, const uint4* __restrict__ wq, unsigned* rep, int cold) {
    code_shift<SHIFT>();
    if (cold) asm volatile("s_icache_inv\n\ts_nop 7\n\ts_nop 7");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int j = tid; j < L_BYTES / 16; j += 512) reinterpret_cast<uint4*>(smem)[j] = wq[j];
    float xin[4][8];
    {
        const float* p = x + ((size_t)blockIdx.x * 256 + wave * 32 + l31) * 64 + 8 * half;      // K step t, lane half h = inputs 16 t + 8 h .. + 7
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 u0 = *reinterpret_cast<const float4*>(p + t * 16), u1 = *reinterpret_cast<const float4*>(p + t * 16 + 4);
            xin[t][0] = u0.x; xin[t][1] = u0.y; xin[t][2] = u0.z; xin[t][3] = u0.w; xin[t][4] = u1.x; xin[t][5] = u1.y; xin[t][6] = u1.z; xin[t][7] = u1.w;
        }
    }
    __syncthreads();
    f32x16 first[2], out[2];
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 16; ++r) out[mb][r] = 0.25f * (float)(mb + 1);
        asm volatile("" ::: "memory");
        bf16x8 w[2][2][3];
        auto ldw = [&](int t, bf16x8 (&o)[2][3]) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int q = 0; q < 3; ++q) o[mb][q] = *reinterpret_cast<const bf16x8*>(smem + (((t * 2 + mb) * 3 + q) * 64 + lane) * 16);
        };
        ldw(0, w[0]);
        bf16x8 xf[2][3];
        {
            float y[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) y[i] = fmaf(xin[0][i], 1.25f, 0.125f);
            split8(y, xf[0][0], xf[0][1], xf[0][2]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t + 1 < 4) ldw(t + 1, w[(t + 1) & 1]);
            asm volatile("" ::: "memory");
            const bf16x8 xh = xf[t & 1][0], xm = xf[t & 1][1], xl = xf[t & 1][2];
            __builtin_amdgcn_sched_barrier(0);
#define MM(WQ, X) { _Pragma("unroll") for (int mb = 0; mb < 2; ++mb) out[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[t & 1][mb][WQ], X, out[mb], 0, 0, 0); }
            MM(2, xh) MM(0, xl) MM(1, xm) MM(1, xh) MM(0, xm) MM(0, xh)
#undef MM
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < 4) {
                float y[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) y[i] = fmaf(xin[t + 1][i], 1.25f, 0.125f);
                split8(y, xf[(t + 1) & 1][0], xf[(t + 1) & 1][1], xf[(t + 1) & 1][2]);
                asm volatile("" : "+v"(xf[(t + 1) & 1][0]), "+v"(xf[(t + 1) & 1][1]), "+v"(xf[(t + 1) & 1][2])
                                : "v"(xf[t & 1][0]), "v"(xf[t & 1][1]), "v"(xf[t & 1][2]), "v"(w[t & 1][0][0]), "v"(w[t & 1][0][1]), "v"(w[t & 1][0][2]),
                                  "v"(w[t & 1][1][0]), "v"(w[t & 1][1][1]), "v"(w[t & 1][1][2]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(out[0]), "+v"(out[1]));
        __builtin_amdgcn_sched_barrier(0);
        if (pass == 0) { first[0] = out[0]; first[1] = out[1]; }      // (wave-uniform branch; a res[pass] array would live in scratch memory)
    }
    unsigned bad = 0;      // bit mb * 16 + r: register r of accumulator mb differs between the cold and the warm pass
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (__float_as_uint(first[mb][r]) != __float_as_uint(out[mb][r])) bad |= 1u << (mb * 16 + r);
    if (bad) {
        const unsigned n = atomicAdd(rep, 1u);
        if (n < 4096) { rep[4 + 4 * n] = blockIdx.x; rep[5 + 4 * n] = (unsigned)wave; rep[6 + 4 * n] = (unsigned)lane; rep[7 + 4 * n] = bad; }
    }
}

template <int S>
static void launch(int shift, const float* x, const uint4* wq, unsigned* rep, int cold, hipStream_t st) {
    if (shift == S) { hipLaunchKernelGGL(refill_kernel<S>, dim3(256), dim3(512), L_BYTES, st, x, wq, rep, cold); return; }
    if constexpr (S < 15) launch<S + 1>(shift, x, wq, rep, cold, st);
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 3.0;
    const size_t nx = (size_t)256 * 256 * 64;
    std::vector<float> hx(nx);
    unsigned s = 12345u;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (float)(s >> 8) * (1.0f / 16777216.0f) * 4.f - 2.f; }
    std::vector<unsigned short> hw(L_BYTES / 2);
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; const float f = ((float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.25f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
    float* dx; uint4* dw; unsigned* rep;
    hipMalloc(&dx, nx * 4); hipMalloc(&dw, L_BYTES); hipMalloc(&rep, (4 + 4 * 4096) * 4);
    hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(dw, hw.data(), L_BYTES, hipMemcpyHostToDevice);
    std::vector<unsigned> h(4 + 4 * 4096);
    for (int cold = 1; cold >= 0; --cold)
        for (int shift = 0; shift < 16; ++shift) {
            hipMemset(rep, 0, (4 + 4 * 4096) * 4);
            const auto t0 = std::chrono::steady_clock::now();
            long n = 0;
            const double budget = cold ? secs : secs * 0.25;
            while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < budget) {
                for (int i = 0; i < 200; ++i) launch<0>(shift, dx, dw, rep, cold, 0);
                hipDeviceSynchronize();
                n += 200;
            }
            hipMemcpy(h.data(), rep, h.size() * 4, hipMemcpyDeviceToHost);
            unsigned grp[4] = {0, 0, 0, 0}, acc[2] = {0, 0}, waves[8] = {0};
            const unsigned m = h[0] < 4096 ? h[0] : 4096;
            for (unsigned k = 0; k < m; ++k) { grp[(h[6 + 4 * k] & 63) >> 4]++; waves[h[5 + 4 * k] & 7]++; if (h[7 + 4 * k] & 0xffffu) acc[0]++; if (h[7 + 4 * k] >> 16) acc[1]++; }
            printf("%s shift %2d: %ld launches (256 workgroups x 8 waves, the layer twice per wave): %u lanes whose first pass differs from their second; lanes 0-15 / 16-31 / 32-47 / 48-63: %u %u %u %u; "
                   "accumulator 0 / 1: %u %u; waves 0..7: %u %u %u %u %u %u %u %u\n", cold ? "COLD" : "warm", shift, n, h[0], grp[0], grp[1], grp[2], grp[3], acc[0], acc[1],
                   waves[0], waves[1], waves[2], waves[3], waves[4], waves[5], waves[6], waves[7]);
            fflush(stdout);
        }
    return 0;
}
