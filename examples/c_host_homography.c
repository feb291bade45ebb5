/* A plain C99 host on the C ABI (include/xfeat_hip.h): no Python, no PyTorch -- HIP runtime calls for memory and the stream,
 * xfh_find_homography for the demo's cv2.findHomography(USAC_MAGSAC) step (realtime_demo.py:223-229 of the reference).
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_host_homography.c \
 *       -L accelerated_features_amd -lxfeat_hip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/accelerated_features_amd -o c_host_homography
 *
 * 600 correspondences under a known similarity transform, a third of them replaced by random points; exit code 0 = the estimate is
 * within 0.05 px of the truth on the image corners and the inlier mask is the expected one. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "xfeat_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_XFH(x) do { int r_ = (x); if (r_ != XFH_OK) { fprintf(stderr, "%s: %d %s\n", #x, r_, xfh_last_error()); return 3; } } while (0)

static uint32_t lcg(uint32_t* s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }
static float uni(uint32_t* s, float hi) { return hi * (float)lcg(s) / 16777216.0f; }

int main(void) {
    enum { N = 600, ITERS = 700 };
    const double Ht[9] = {0.96, -0.10, 31.0, 0.10, 0.96, -12.0, 0.0, 0.0, 1.0};
    static float p0[N][2], p1[N][2];
    static uint8_t expect[N];
    uint32_t s = 12345u;
    for (int i = 0; i < N; ++i) {
        p0[i][0] = uni(&s, 640.f); p0[i][1] = uni(&s, 480.f);
        if (i % 3 == 2) { p1[i][0] = uni(&s, 640.f); p1[i][1] = uni(&s, 480.f); }
        else { p1[i][0] = (float)(Ht[0] * p0[i][0] + Ht[1] * p0[i][1] + Ht[2]); p1[i][1] = (float)(Ht[3] * p0[i][0] + Ht[4] * p0[i][1] + Ht[5]); }
        const double dx = Ht[0] * p0[i][0] + Ht[1] * p0[i][1] + Ht[2] - p1[i][0], dy = Ht[3] * p0[i][0] + Ht[4] * p0[i][1] + Ht[5] - p1[i][1];
        expect[i] = dx * dx + dy * dy < 16.0;
    }
    hipStream_t st;
    float *d0, *d1;
    double* dH;
    uint8_t* dmask;
    int32_t* dinfo;
    void* ws;
    const size_t wsb = xfh_homography_workspace_bytes(1, ITERS);
    CHECK_HIP(hipStreamCreate(&st));
    CHECK_HIP(hipMalloc((void**)&d0, sizeof p0)); CHECK_HIP(hipMalloc((void**)&d1, sizeof p1));
    CHECK_HIP(hipMalloc((void**)&dH, 9 * sizeof(double))); CHECK_HIP(hipMalloc((void**)&dmask, N)); CHECK_HIP(hipMalloc((void**)&dinfo, 8 * sizeof(int32_t)));
    CHECK_HIP(hipMalloc(&ws, wsb));
    CHECK_HIP(hipMemcpyAsync(d0, p0, sizeof p0, hipMemcpyHostToDevice, st));
    CHECK_HIP(hipMemcpyAsync(d1, p1, sizeof p1, hipMemcpyHostToDevice, st));
    CHECK_XFH(xfh_find_homography(d0, d1, NULL, N, 1, N, 4.0, ITERS, 0.995, 0, dH, dmask, dinfo, ws, wsb, st));
    double H[9];
    static uint8_t mask[N];
    int32_t info[8];
    CHECK_HIP(hipMemcpyAsync(H, dH, sizeof H, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipMemcpyAsync(mask, dmask, N, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipMemcpyAsync(info, dinfo, sizeof info, hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipStreamSynchronize(st));
    double worst = 0.0;
    const double corners[4][2] = {{0, 0}, {640, 0}, {0, 480}, {640, 480}};
    for (int c = 0; c < 4; ++c) {
        const double x = corners[c][0], y = corners[c][1], w = H[6] * x + H[7] * y + H[8];
        const double ex = (H[0] * x + H[1] * y + H[2]) / w - (Ht[0] * x + Ht[1] * y + Ht[2]), ey = (H[3] * x + H[4] * y + H[5]) / w - (Ht[3] * x + Ht[4] * y + Ht[5]);
        worst = fmax(worst, sqrt(ex * ex + ey * ey));
    }
    int wrong = 0;
    for (int i = 0; i < N; ++i) wrong += mask[i] != expect[i];
    printf("found %d, winner %d after %d iterations, %d inliers, corner error %.2e px, mask differences %d\n", info[0], info[1], info[2], info[3], worst, wrong);
    hipFree(d0); hipFree(d1); hipFree(dH); hipFree(dmask); hipFree(dinfo); hipFree(ws); hipStreamDestroy(st);
    return info[0] == 1 && worst < 0.05 && wrong == 0 ? 0 : 1;
}
