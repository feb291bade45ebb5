// Store-path microbenchmark: how long does a workgroup need to ISSUE its output stores, by access pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// out: [B][C][H][W] floats; each workgroup (8 waves) writes a 16x16-pixel region x 64 channels like the conv epilogues.
template <int PAT>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int H, int W, int tiles_x, int tiles) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x / tiles, tile = blockIdx.x % tiles;
    const int oy0 = (tile / tiles_x) * 16, ox0 = (tile % tiles_x) * 16;
    const size_t HW = (size_t)H * W;
    float* ob = out + (size_t)b * 64 * HW;
    const float v = tid * 0.5f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (PAT == 0) {          // Winograd epilogue: wave -> (row parity oi, cout block, tile block); lane = tile (8x8 tiles of 2x2 px), float2
        const int oi = wave & 1, blk = (wave >> 1) & 1, tb = wave >> 2;
        const int t = tb * 32 + l31, ty = t >> 3, tx = t & 7;
        float* o = ob + (size_t)(oy0 + 2 * ty + oi) * W + ox0 + 2 * tx;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            *reinterpret_cast<float2*>(o + co * HW) = make_float2(v + r, v - r);
        }
    } else if (PAT == 1) {   // direct-kernel epilogue: wave -> 64 consecutive region pixels (2 blocks of 32), dword stores, 2x16 couts per block
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int t = (wave * 2 + n) * 32 + l31 - (wave >= 4 ? 256 : 0);
            const int ty = (t >> 4) & 15, tx = t & 15;
            float* o = ob + (size_t)(oy0 + ty) * W + ox0 + tx;
#pragma unroll
            for (int m = 0; m < 1; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = (wave >= 4 ? 32 : 0) + (r & 3) + 8 * (r >> 2) + 4 * half;
                    o[co * HW] = v + r;
                }
        }
    } else if (PAT == 2) {   // same bytes, one 16-pixel row x 4 channels per instruction (dword), channel-major lanes
        // wave handles 8 channels; instr i: rows 4i..4i+3?  lanes: 16 px x 4 rows
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ty = q * 4 + (lane >> 4), tx = lane & 15;
                ob[(size_t)(wave * 8 + c) * HW + (size_t)(oy0 + ty) * W + ox0 + tx] = v + c;
            }
    } else if (PAT == 3) {   // float4 per lane: 4 lanes = one 16-px row; instr covers 16 rows of one channel
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int ty = lane >> 2, tx = (lane & 3) * 4;
            *reinterpret_cast<float4*>(ob + (size_t)(wave * 8 + c) * HW + (size_t)(oy0 + ty) * W + ox0 + tx) = make_float4(v, v + 1, v + 2, v + c);
        }
    } else {                 // fully contiguous: 1 KiB per instruction (float4, lanes consecutive) -- upper bound
#pragma unroll
        for (int c = 0; c < 8; ++c)
            *reinterpret_cast<float4*>(out + ((size_t)blockIdx.x * 64 + wave * 8 + c) * 256 + lane * 4) = make_float4(v, v + 1, v + 2, v + c);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int PAT>
void run(const char* name, int B) {
    const int H = 64, W = 80, tiles_x = 5, tiles = 20;
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * B * 64 * H * W + (1 << 20));
    hipMalloc(&cyc, sizeof(long long) * B * tiles * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) k<PAT><<<B * tiles, 512>>>(out, cyc, H, W, tiles_x, tiles);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<PAT><<<B * tiles, 512>>>(out, cyc, H, W, tiles_x, tiles);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(B * tiles * 8); hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    double mean = 0, mx = 0; for (auto x : h) { mean += x; mx = x > mx ? x : mx; } mean /= h.size();
    printf("%-34s B=%2d: issue cycles per wave mean %7.0f max %7.0f | kernel %.1f us (%.2f TB/s)\n", name, B, mean, mx, ms * 1e3,
           (double)B * 64 * H * W * 4 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int B : {3, 64}) {
        run<0>("winograd float2 (8x8 tiles)", B);
        run<1>("direct dword, 32 px per half-wave", B);
        run<2>("dword, 4 rows x 16 px per instr", B);
        run<3>("float4, 16 rows x 16 px per instr", B);
        run<4>("contiguous float4 (upper bound)", B);
    }
    return 0;
}
