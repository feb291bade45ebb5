// Two assumptions the kernels prepared on CPU make about the hardware, checked in a few milliseconds (first step of tools/gpu_r5_first.sh):
//  1. LDS-DMA of 16 bytes per lane under a PARTIAL exec mask (block1 mode 7: the last piece of conv3's weight image; conv_bx64 SP: the last piece of a chunk tile, the rows
//     a half tile does not need): an active lane writes LDS base + 16 * lane -- its own slot, not a compacted one -- and an inactive lane writes nothing.
//  2. the operand layout of v_mfma_f32_16x16x32_f16: lane l holds A[row l & 15][k = 8 (l >> 4) .. + 7] and B[k][column l & 15], D[4 (l >> 4) + j][l & 15] in register j.
//     /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/bench_src/hw_semantics.hip -o /tmp/hw_semantics && /tmp/hw_semantics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void dma_partial(const unsigned* src, unsigned* out, unsigned long long mask) {
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4];
    const int lane = threadIdx.x;
    for (int i = 0; i < 4; ++i) lds[lane * 4 + i] = 0xdeadbeefu;
    __syncthreads();
    if ((mask >> lane) & 1) __builtin_amdgcn_global_load_lds((gptr_t)(src + lane * 4), (lptr_t)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[lane * 4 + i];
}

__global__ void mfma_layout(const _Float16* A /* 16 x 32 */, const _Float16* B /* 32 x 16 */, float* D /* 16 x 16 */) {
    const int l = threadIdx.x;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j]; b[j] = B[(8 * (l >> 4) + j) * 16 + (l & 15)]; }
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
    for (int j = 0; j < 4; ++j) D[(4 * (l >> 4) + j) * 16 + (l & 15)] = d[j];
}

__global__ __launch_bounds__(512) void occ_probe(float* p) {      // (a small kernel of 512 threads: what limits its residency is the dynamic LDS it is launched with)
    extern __shared__ float sm[];
    sm[threadIdx.x] = p[threadIdx.x];
    __syncthreads();
    p[threadIdx.x] = sm[511 - threadIdx.x];
}

int main() {
    int bad = 0;
    // 3. how many 512-thread workgroups with N bytes of dynamic LDS the runtime places on a CU: block1 needs three with 51 728 (mode 5), 51 216 (mode 6), 53 616 (mode 7);
    //    160 KB / 3 = 54 613, less whatever granule LDS is handed out in
    for (int bytes : {51216, 51728, 52480, 53616, 53760, 54128, 54272, 54613, 55296}) {
        int n = -1;
        hipFuncSetAttribute(reinterpret_cast<const void*>(occ_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, occ_probe, 512, bytes);
        printf("512 threads + %d bytes of LDS: %d workgroups per CU\n", bytes, n);
    }
    unsigned *src, *out;
    hipMalloc(&src, 1024); hipMalloc(&out, 1024);
    std::vector<unsigned> hs(256), ho(256);
    for (int i = 0; i < 256; ++i) hs[i] = 0x1000u + i;
    hipMemcpy(src, hs.data(), 1024, hipMemcpyHostToDevice);
    const unsigned long long masks[4] = {0xfffffull, 0x3fffffull, 0x9249249249249249ull, 0xffffffff00000000ull};
    for (unsigned long long m : masks) {
        dma_partial<<<1, 64>>>(src, out, m);
        hipMemcpy(ho.data(), out, 1024, hipMemcpyDeviceToHost);
        int wrong = 0;
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) wrong += ho[l * 4 + i] != (((m >> l) & 1) ? hs[l * 4 + i] : 0xdeadbeefu);
        printf("LDS-DMA b128, exec mask %016llx: %s (%d words differ from 'own slot, inactive lanes untouched')\n", m, wrong ? "DIFFERENT" : "as assumed", wrong);
        bad += wrong != 0;
    }
    _Float16 *A, *B; float* D;
    hipMalloc(&A, 16 * 32 * 2); hipMalloc(&B, 32 * 16 * 2); hipMalloc(&D, 256 * 4);
    std::vector<_Float16> ha(512), hb(512); std::vector<float> hd(256);
    srand(1);
    for (int i = 0; i < 512; ++i) { ha[i] = (_Float16)((rand() % 17 - 8) / 8.0f); hb[i] = (_Float16)((rand() % 17 - 8) / 8.0f); }
    hipMemcpy(A, ha.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(B, hb.data(), 1024, hipMemcpyHostToDevice);
    mfma_layout<<<1, 64>>>(A, B, D);
    hipMemcpy(hd.data(), D, 1024, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            double s = 0;
            for (int k = 0; k < 32; ++k) s += (double)(float)ha[m * 32 + k] * (double)(float)hb[k * 16 + n];
            worst = fmax(worst, fabs(s - hd[m * 16 + n]));
        }
    printf("v_mfma_f32_16x16x32_f16 operand layout: %s (max |D - A B| = %g)\n", worst < 1e-5 ? "as assumed" : "DIFFERENT", worst);
    bad += !(worst < 1e-5);
    printf(bad ? "RESULT: %d assumption(s) do not hold\n" : "RESULT: both assumptions hold\n", bad);
    return bad ? 1 : 0;
}
