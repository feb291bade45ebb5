// Where a unit of conv_bx64s2x_kernel (block4.0 / block5.0) spends its time: s_memtime stamps of wave 0 in the workgroup's SECOND unit (xfh_debug_trace; conv_bx64s2_body.hpp
// S2_STAMP: [0] unit start; tap row r: [1 + 4 r] in front of the wait, [2 + 4 r] barrier passed, [3 + 4 r] its MFMAs issued; [50] output stores issued), one layer through xfh_conv_layer.
//   build (CPU):  hipcc -O2 -w --offload-arch=gfx950 tools/bench_src/s2_trace_probe.cpp -o gpurun_probe/s2_trace_probe -ldl
//   run (GPU):    gpurun_probe/s2_trace_probe accelerated_features_amd/libxfeat_hip.so gpurun_probe/weights.bin [layer 10] [B 64] [H 60] [W 80]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(3); } } while (0)
typedef void* H;

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    if (argc < 3) { printf("usage: s2_trace_probe <libxfeat_hip.so> <weights.bin> [layer] [B] [H] [W]\n"); return 1; }
    const int layer = argc > 3 ? atoi(argv[3]) : 10, B = argc > 4 ? atoi(argv[4]) : 64, Hh = argc > 5 ? atoi(argv[5]) : 60, W = argc > 6 ? atoi(argv[6]) : 80;
    void* so = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!so) { printf("dlopen: %s\n", dlerror()); return 2; }
#define SYM(T, name) auto name = reinterpret_cast<T>(dlsym(so, #name)); if (!name) { printf("missing %s\n", #name); return 2; }
    SYM(int (*)(const float* const*, int, int, H*), xfh_create)
    SYM(const char* (*)(), xfh_last_error)
    SYM(int (*)(H, int, const float*, int, int, int, float*, int, void*), xfh_conv_layer)
    SYM(int (*)(H, long long*), xfh_debug_trace)
    FILE* f = fopen(argv[2], "rb");
    if (!f) { printf("cannot open %s\n", argv[2]); return 2; }
    int na = 0;
    if (fread(&na, 4, 1, f) != 1) return 2;
    std::vector<std::vector<float>> arrs(na);
    std::vector<const float*> ptrs(na);
    for (int i = 0; i < na; ++i) { int n; if (fread(&n, 4, 1, f) != 1) return 2; arrs[i].resize(n); if (fread(arrs[i].data(), 4, n, f) != (size_t)n) return 2; ptrs[i] = arrs[i].data(); }
    fclose(f);
    H h = nullptr;
    if (xfh_create(ptrs.data(), na, 0, &h)) { printf("xfh_create: %s\n", xfh_last_error()); return 2; }
    const size_t n = (size_t)B * 64 * Hh * W;
    std::vector<float> hx(n);
    const size_t ny = (size_t)B * 128 * Hh * W;
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; hx[i] = ((s >> 8) * (1.0f / 16777216.0f)); }
    float *x, *y;
    long long* tr;
    const int NWG = 256;
    HIPCHK(hipMalloc(&x, n * 4)); HIPCHK(hipMalloc(&y, ny * 4)); HIPCHK(hipMalloc(&tr, (size_t)NWG * 64 * 8));
    HIPCHK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int pass = 0; pass < 2; ++pass) {        // 0: production kernel timed with events; 1: the stamped twin
        xfh_debug_trace(h, pass ? tr : nullptr);
        for (int i = 0; i < 5; ++i) if (xfh_conv_layer(h, layer, x, B, Hh, W, y, 2, nullptr)) { printf("xfh_conv_layer: %s\n", xfh_last_error()); return 2; }
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemset(tr, 0, (size_t)NWG * 64 * 8));
        HIPCHK(hipEventRecord(e0, nullptr));
        const int reps = pass ? 1 : 20;
        for (int i = 0; i < reps; ++i) xfh_conv_layer(h, layer, x, B, Hh, W, y, 2, nullptr);
        HIPCHK(hipEventRecord(e1, nullptr));
        HIPCHK(hipDeviceSynchronize());
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.2f us per launch (layer %d, B %d, %d x %d)\n", pass ? "stamped twin" : "production kernel", 1e3 * ms / reps, layer, B, Hh, W);
    }
    std::vector<long long> t((size_t)NWG * 64);
    HIPCHK(hipMemcpy(t.data(), tr, t.size() * 8, hipMemcpyDeviceToHost));
    printf("s_memtime counts (~0.77 of a shader cycle here: three units + prologue = the launch) of wave 0's second unit, per row: wait+barrier / stamp 2 -> stamp 3 / stamp 3 -> the next row's stamp 1\n"
           "(the multiplying waves open row r + 1 in front of row r's last MFMAs: stamp 3 of a row lies behind stamp 2 of the next, the third figure is negative;\n"
           " barrier to barrier = second + third + the next row's first)\n");
    std::vector<long long> unit;
    std::vector<std::vector<long long>> wt(12), mf(12), gap(12);
    for (int g = 0; g < NWG; ++g) {
        const long long* q = &t[g * 64];
        if (!q[0] || !q[50]) continue;
        unit.push_back(q[50] - q[0]);
        for (int r = 0; r < 12; ++r) {
            wt[r].push_back(q[2 + 4 * r] - q[1 + 4 * r]);
            mf[r].push_back(q[3 + 4 * r] - q[2 + 4 * r]);
            gap[r].push_back((r < 11 ? q[1 + 4 * (r + 1)] : q[50]) - q[3 + 4 * r]);
        }
        if (g < 6 || g % 61 == 0) {
            printf("wg %3d unit %5lld:", g, q[50] - q[0]);
            for (int r = 0; r < 12; ++r) printf("  %lld/%lld/%lld", q[2 + 4 * r] - q[1 + 4 * r], q[3 + 4 * r] - q[2 + 4 * r], (r < 11 ? q[1 + 4 * (r + 1)] : q[50]) - q[3 + 4 * r]);
            printf("   first stamp -> row 0: %lld\n", q[1] - q[0]);
        }
    }
    auto med = [](std::vector<long long>& v) { if (v.empty()) return -1ll; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("workgroups with a second unit: %zu; unit (start -> stores issued) median %lld counts\n", unit.size(), med(unit));
    long long sw = 0, sm = 0, sg = 0;
    for (int r = 0; r < 12; ++r) { const long long a = med(wt[r]), b = med(mf[r]), c = med(gap[r]); sw += a; sm += b; sg += c; printf("  row %2d (chunk %d, dy %d): wait+barrier %4lld   MFMA block %4lld   tail %4lld\n", r, r / 3, r % 3, a, b, c); }
    printf("  sum of medians: wait+barrier %lld   MFMA blocks %lld   tails %lld\n", sw, sm, sg);
    return 0;
}
