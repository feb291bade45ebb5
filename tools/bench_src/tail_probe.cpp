// Per-kernel HIP-event times of the backbone (B x 3 x H x W random images) and of the matcher (P pairs of N random unit descriptors) through the C ABI alone -- no Python:
// a gpurun call with this prebuilt binary is charged ~20 s.  Prints one line per span of XFH_PROF_ALL (us per launch, averaged), the sum, and the sum of the small-map tail
// (block4.*, block5.*, pyramid) that round 6 works on; optional parity of the backbone's feats against a second handle on the fp32-range kernels (fx = 0, block1 = 5).
//   build (CPU):  python tools/dump_weights.py gpurun_probe/weights.bin
//                 hipcc -O2 -w --offload-arch=gfx950 tools/bench_src/tail_probe.cpp -o gpurun_probe/tail_probe -ldl
//   run (GPU):    gpurun_probe/tail_probe accelerated_features_amd/libxfeat_hip.so gpurun_probe/weights.bin [iters 20] [B 64] [H 480] [W 640] [check 1]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(3); } } while (0)
typedef void* H;

static std::vector<float> rnd(size_t n, unsigned seed, float lo, float hi) {
    std::vector<float> v(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; v[i] = lo + (hi - lo) * ((s >> 8) * (1.0f / 16777216.0f)); }
    return v;
}

static const char* span_name(int id) {
    static const char* conv[] = {"skip1", "block1.0", "block1.1", "block1.2", "block1.3", "block2.0", "block2.1", "block3.0", "block3.1(+.2)", "block3.2", "block4.0", "block4.1", "block4.2",
                                 "block5.0", "block5.1", "block5.2", "block5.3 (own launch)", "block_fusion.0", "block_fusion.1(+.2)", "block_fusion.2"};
    switch (id) {
        case 3: return "block1";
        case 200: return "gray_stats + coef";
        case 201: return "pyramid (+ block5.3)";
        case 202: return "head rel";
        case 203: return "head kp";
        case 220: return "match memset";
        case 221: return "match prep";
        case 222: return "match sweep";
        case 223: return "match thr + refine";
        case 224: return "match finalize";
        case 225: return "match exact";
    }
    if (id >= 100 && id < 120) return conv[id - 100];
    return "?";
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    if (argc < 3) { printf("usage: tail_probe <libxfeat_hip.so> <weights.bin> [iters] [B] [H] [W] [check]\n"); return 1; }
    const int iters = argc > 3 ? atoi(argv[3]) : 20, B = argc > 4 ? atoi(argv[4]) : 64, Hh = argc > 5 ? atoi(argv[5]) : 480, W = argc > 6 ? atoi(argv[6]) : 640;
    const bool check = argc > 7 ? atoi(argv[7]) != 0 : true;
    void* so = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!so) { printf("dlopen: %s\n", dlerror()); return 2; }
#define SYM(T, name) auto name = reinterpret_cast<T>(dlsym(so, #name)); if (!name) { printf("missing %s\n", #name); return 2; }
    SYM(int (*)(const float* const*, int, int, H*), xfh_create)
    SYM(const char* (*)(), xfh_last_error)
    SYM(size_t (*)(int, int, int, int), xfh_backbone_workspace_bytes)
    SYM(int (*)(H, const float*, int, int, int, int, float*, float*, float*, float*, float*, void*, size_t, void*), xfh_backbone)
    SYM(size_t (*)(int, int, int), xfh_match_workspace_bytes)
    SYM(int (*)(H, const float*, size_t, const float*, size_t, const uint16_t*, const uint16_t*, const int32_t*, const int32_t*, int, int, int, int, int, float, int64_t*, int64_t*, int32_t*, void*, size_t, void*), xfh_match_mnn)
    SYM(int (*)(H, int), xfh_profile_select)
    SYM(int (*)(H, int*, double*, int, int*), xfh_profile_read_spans)
    SYM(int (*)(H, const char*, int), xfh_set_option)
    SYM(int (*)(H, int32_t*), xfh_set_status_buffer)
    FILE* f = fopen(argv[2], "rb");
    if (!f) { printf("cannot open %s\n", argv[2]); return 2; }
    int na = 0;
    if (fread(&na, 4, 1, f) != 1) return 2;
    std::vector<std::vector<float>> arrs(na);
    std::vector<const float*> ptrs(na);
    for (int i = 0; i < na; ++i) { int n; if (fread(&n, 4, 1, f) != 1) return 2; arrs[i].resize(n); if (fread(arrs[i].data(), 4, n, f) != (size_t)n) return 2; ptrs[i] = arrs[i].data(); }
    fclose(f);
    H h = nullptr, h32 = nullptr;
    if (xfh_create(ptrs.data(), na, 0, &h)) { printf("xfh_create: %s\n", xfh_last_error()); return 2; }
    int32_t* status;
    HIPCHK(hipMalloc(&status, 4)); HIPCHK(hipMemset(status, 0, 4));
    xfh_set_status_buffer(h, status);

    const size_t npx = (size_t)B * Hh * W, ncell = npx / 64;
    // images with structure at every scale (so that the activations are not noise-flat): a sum of three square waves + noise
    std::vector<float> himg(3 * npx);
    {
        auto nz = rnd(3 * npx, 7, 0.f, 0.25f);
        for (int b = 0; b < B; ++b)
            for (int c = 0; c < 3; ++c)
                for (int y = 0; y < Hh; ++y)
                    for (int x = 0; x < W; ++x) {
                        const size_t i = (((size_t)b * 3 + c) * Hh + y) * W + x;
                        himg[i] = 0.25f * (((x + 3 * b) / 7 + y / 5) & 1) + 0.25f * (((x / 31) + (y + b) / 23) & 1) + 0.25f * (((x + y) / 61) & 1) + nz[i];
                    }
    }
    float *img, *feats, *heat, *rel, *inv, *feats32 = nullptr;
    HIPCHK(hipMalloc(&img, 3 * npx * 4)); HIPCHK(hipMalloc(&feats, ncell * 64 * 4)); HIPCHK(hipMalloc(&heat, npx * 4)); HIPCHK(hipMalloc(&rel, ncell * 4)); HIPCHK(hipMalloc(&inv, ncell * 4));
    HIPCHK(hipMemcpy(img, himg.data(), 3 * npx * 4, hipMemcpyHostToDevice));
    const size_t wsb = xfh_backbone_workspace_bytes(B, 3, Hh, W);
    void* ws;
    HIPCHK(hipMalloc(&ws, wsb));
    auto backbone = [&](H hh, float* fo) {
        if (xfh_backbone(hh, img, B, 3, Hh, W, fo, nullptr, heat, rel, inv, ws, wsb, nullptr)) { printf("xfh_backbone: %s\n", xfh_last_error()); exit(2); }
    };
    // matcher inputs: P pairs of N unit rows, the second set a noisy permutation of the first (so that there are mutual matches)
    const int P = B / 2 > 0 ? B / 2 : 1, N = 4096;
    std::vector<float> d1((size_t)P * N * 64), d2((size_t)P * N * 64);
    {
        auto r = rnd((size_t)P * N * 64, 11, -1.f, 1.f), e = rnd((size_t)P * N * 64, 13, -0.15f, 0.15f);
        for (size_t row = 0; row < (size_t)P * N; ++row) {
            double s = 0;
            for (int k = 0; k < 64; ++k) s += (double)r[row * 64 + k] * r[row * 64 + k];
            const float is = (float)(1.0 / std::sqrt(s));
            for (int k = 0; k < 64; ++k) d1[row * 64 + k] = r[row * 64 + k] * is;
            const size_t p = row / N, j = (row % N * 2654435761u) % N, dst = p * N + j;
            double s2 = 0;
            for (int k = 0; k < 64; ++k) { const float v = d1[row * 64 + k] + e[row * 64 + k]; d2[dst * 64 + k] = v; s2 += (double)v * v; }
            const float is2 = (float)(1.0 / std::sqrt(s2));
            for (int k = 0; k < 64; ++k) d2[dst * 64 + k] *= is2;
        }
    }
    float *gd1, *gd2; int64_t *i0, *i1; int32_t *nm, *nv;
    HIPCHK(hipMalloc(&gd1, d1.size() * 4)); HIPCHK(hipMalloc(&gd2, d2.size() * 4)); HIPCHK(hipMalloc(&i0, (size_t)P * N * 8)); HIPCHK(hipMalloc(&i1, (size_t)P * N * 8));
    HIPCHK(hipMalloc(&nm, P * 4)); HIPCHK(hipMalloc(&nv, 2 * P * 4));
    HIPCHK(hipMemcpy(gd1, d1.data(), d1.size() * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(gd2, d2.data(), d2.size() * 4, hipMemcpyHostToDevice));
    { std::vector<int32_t> n(2 * P, N); HIPCHK(hipMemcpy(nv, n.data(), 2 * P * 4, hipMemcpyHostToDevice)); }
    const size_t mwsb = xfh_match_workspace_bytes(P, N, N);
    void* mws;
    HIPCHK(hipMalloc(&mws, mwsb));
    auto match = [&] {
        if (xfh_match_mnn(h, gd1, (size_t)N * 64, gd2, (size_t)N * 64, nullptr, nullptr, nv, nv, 1, P, P, N, N, 0.82f, i0, i1, nm, mws, mwsb, nullptr)) { printf("xfh_match_mnn: %s\n", xfh_last_error()); exit(2); }
    };

    for (int i = 0; i < 3; ++i) { backbone(h, feats); match(); }
    HIPCHK(hipDeviceSynchronize());
    xfh_profile_select(h, 1000);
    std::map<int, std::pair<double, int>> acc;      // span id -> (ms, launches)
    std::vector<int> order;
    for (int it = 0; it < iters; ++it) {
        backbone(h, feats);
        match();
        HIPCHK(hipDeviceSynchronize());
        int ids[128], n = 0;
        double ms[128];
        xfh_profile_read_spans(h, ids, ms, 128, &n);
        for (int i = 0; i < n && i < 128; ++i) {
            if (!acc.count(ids[i])) order.push_back(ids[i]);
            acc[ids[i]].first += ms[i]; acc[ids[i]].second += 1;
        }
    }
    xfh_profile_select(h, 0);
    double tot = 0, tail = 0, mt = 0;
    printf("B %d  %d x %d  (%d iterations; HIP-event spans, one launch stream)\n", B, Hh, W, iters);
    for (int id : order) {
        const double us = 1e3 * acc[id].first / iters;
        printf("  %-22s %8.1f us per step  (%d launch%s)\n", span_name(id), us, acc[id].second / iters, acc[id].second / iters == 1 ? "" : "es");
        tot += us;
        if ((id >= 110 && id <= 116) || id == 201) tail += us;
        if (id >= 220) mt += us;
    }
    printf("sum %.1f us   small-map tail (block4.*, block5.*, pyramid) %.1f us   matcher %.1f us\n", tot, tail, mt);
    { std::vector<int32_t> n(P); HIPCHK(hipMemcpy(n.data(), nm, P * 4, hipMemcpyDeviceToHost)); long s = 0; for (int v : n) s += v; printf("matches per pair: %.1f\n", (double)s / P); }
    int32_t st = 0;
    HIPCHK(hipMemcpy(&st, status, 4, hipMemcpyDeviceToHost));
    printf("status word %d\n", st);
    if (check) {
        if (xfh_create(ptrs.data(), na, 0, &h32)) { printf("xfh_create: %s\n", xfh_last_error()); return 2; }
        xfh_set_option(h32, "fx", 0); xfh_set_option(h32, "block1", 5);
        HIPCHK(hipMalloc(&feats32, ncell * 64 * 4));
        backbone(h, feats);
        backbone(h32, feats32);
        HIPCHK(hipDeviceSynchronize());
        std::vector<float> a(ncell * 64), b(ncell * 64);
        HIPCHK(hipMemcpy(a.data(), feats, a.size() * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(b.data(), feats32, b.size() * 4, hipMemcpyDeviceToHost));
        double d = 0, m = 0; size_t bad = 0;
        for (size_t i = 0; i < a.size(); ++i) { if (!std::isfinite(a[i])) ++bad; d = std::fmax(d, std::fabs((double)a[i] - b[i])); m = std::fmax(m, std::fabs((double)b[i])); }
        printf("feats: default kernels vs fp32-range kernels: max |diff| %.3g (max |feats| %.3g), non-finite %zu\n", d, m, bad);
    }
    return 0;
}
