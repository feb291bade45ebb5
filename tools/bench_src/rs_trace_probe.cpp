// Where a conv_rs64_kernel launch spends its time on a SMALL map: the stamped twin (xfh_debug_trace: s_memtime of wave 0 at entry, weights requested, first ring
// filled, per-slot stamps of the second unit, exit) of one layer through xfh_conv_layer, printed per workgroup as offsets from the earliest entry of the launch.
//   build (CPU):  hipcc -O2 -w --offload-arch=gfx950 tools/bench_src/rs_trace_probe.cpp -o gpurun_probe/rs_trace_probe -ldl
//   run (GPU):    gpurun_probe/rs_trace_probe accelerated_features_amd/libxfeat_hip.so gpurun_probe/weights.bin [layer 11] [B 64] [H 30] [W 40]
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
    if (argc < 3) { printf("usage: rs_trace_probe <libxfeat_hip.so> <weights.bin> [layer] [B] [H] [W]\n"); return 1; }
    const int layer = argc > 3 ? atoi(argv[3]) : 11, B = argc > 4 ? atoi(argv[4]) : 64, Hh = argc > 5 ? atoi(argv[5]) : 30, W = argc > 6 ? atoi(argv[6]) : 40;
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
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; hx[i] = ((s >> 8) * (1.0f / 16777216.0f)); }
    float *x, *y;
    long long* tr;
    const int NWG = 256;
    HIPCHK(hipMalloc(&x, n * 4)); HIPCHK(hipMalloc(&y, n * 4)); HIPCHK(hipMalloc(&tr, (size_t)NWG * 128 * 8));
    HIPCHK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int pass = 0; pass < 2; ++pass) {        // 0: production kernel timed with events; 1: the stamped twin
        xfh_debug_trace(h, pass ? tr : nullptr);
        for (int i = 0; i < 5; ++i) if (xfh_conv_layer(h, layer, x, B, Hh, W, y, 2, nullptr)) { printf("xfh_conv_layer: %s\n", xfh_last_error()); return 2; }
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemset(tr, 0, (size_t)NWG * 128 * 8));
        HIPCHK(hipEventRecord(e0, nullptr));
        const int reps = pass ? 1 : 20;
        for (int i = 0; i < reps; ++i) xfh_conv_layer(h, layer, x, B, Hh, W, y, 2, nullptr);
        HIPCHK(hipEventRecord(e1, nullptr));
        HIPCHK(hipDeviceSynchronize());
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.2f us per launch (layer %d, B %d, %d x %d)\n", pass ? "stamped twin" : "production kernel", 1e3 * ms / reps, layer, B, Hh, W);
    }
    std::vector<long long> t((size_t)NWG * 128);
    HIPCHK(hipMemcpy(t.data(), tr, t.size() * 8, hipMemcpyDeviceToHost));
    long long tmin = -1, tmax = 0;
    for (int g = 0; g < NWG; ++g) if (t[g * 128]) { if (tmin < 0 || t[g * 128] < tmin) tmin = t[g * 128]; tmax = std::max(tmax, t[g * 128 + 13]); }
    printf("s_memtime counts (100 MHz): launch span entry(min) -> exit(max) = %lld counts\n", tmax - tmin);
    printf("  wg   entry  w_req  ring1   u2start u2_b1t2 u2_bar1 u2_pass u2_b1end u2_b2t2 u2_bar2 u2_pass u2_b2end u2_end   exit  units\n");
    std::vector<long long> ent, pro, ex, per;
    for (int g = 0; g < NWG; ++g) {
        const long long* q = &t[g * 128];
        if (!q[0]) continue;
        ent.push_back(q[0] - tmin); pro.push_back(q[2] - q[0]); ex.push_back(q[13] - tmin);
        if (q[14] > 0) per.push_back((q[13] - q[2]) / q[14]);
        if (g < 12 || g % 37 == 0 || g > NWG - 4) {
            printf("%4d %7lld %6lld %6lld  ", g, q[0] - tmin, q[1] - q[0], q[2] - q[0]);
            for (int k = 3; k <= 12; ++k) printf(" %7lld", q[k] ? q[k] - q[0] : -1);
            printf(" %7lld %5lld\n", q[13] - tmin, q[14]);
        }
    }
    auto stat = [](std::vector<long long>& v, const char* name) {
        if (v.empty()) return;
        std::sort(v.begin(), v.end());
        printf("%-34s min %6lld  median %6lld  max %6lld\n", name, v.front(), v[v.size() / 2], v.back());
    };
    stat(ent, "entry after the first entry"); stat(pro, "prologue (entry -> ring filled)"); stat(per, "counts per unit behind the prologue"); stat(ex, "exit after the first entry");
    return 0;
}
