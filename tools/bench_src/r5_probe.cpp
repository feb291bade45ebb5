// Stand-alone probe of the kernel forms prepared on CPU (no Python, no torch: a gpurun call with a prebuilt binary costs ~15 s of GPU budget).  Loads a build of
// libxfeat_hip.so through its C ABI, creates a handle from a weight dump (gpurun_probe/weights.bin: the arrays of XFeatModel.weight_arrays() of the synthetic test
// weights), and on random VGA frames (B = 64):
//   block1   xfh_debug_block1 in modes 5 / 6 / 7: each against mode 5 (max |diff|), HIP-event time per launch
//   heads    xfh_debug_head_soak variants 101 (default f32 heads), 100, 0 (bf16 split), 102 / 103 / 104 (fp16-pair forms): heat map against 101, time per launch
//   backbone xfh_backbone under option sets: feats / heat / reliability against the default set, time per call
//   scan     (argv[3] = launches per position, with the --scan build) cold-start position scan of the bf16 head (1000 + s), the fp16-pair head (4000 + s) and its
//            LDS round-trip form (5000 + s): launches whose heat map differs from the quiet one
//     hipcc -O2 -w tools/bench_src/r5_probe.cpp -o gpurun_probe/r5_probe -ldl ; gpurun_probe/r5_probe <lib> <weights.bin> [scan launches]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(3); } } while (0)
typedef void* H;
static void* lib;
template <typename F> static F sym(const char* n) { void* p = dlsym(lib, n); if (!p) { printf("missing symbol %s\n", n); exit(2); } return reinterpret_cast<F>(p); }

static std::vector<float> rnd(size_t n, unsigned seed, float lo, float hi) {
    std::vector<float> v(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; v[i] = lo + (hi - lo) * ((s >> 8) * (1.0f / 16777216.0f)); }
    return v;
}
static double maxdiff(const std::vector<float>& a, const std::vector<float>& b, double* amax = nullptr, size_t* nbad = nullptr, double tol = 0) {
    double d = 0, m = 0; size_t nb = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        const double x = std::fabs((double)a[i] - b[i]);
        if (!(x <= d)) d = x;                      // (NaN propagates)
        if (!(x <= tol)) ++nb;
        m = std::fmax(m, std::fabs((double)b[i]));
    }
    if (amax) *amax = m;
    if (nbad) *nbad = nb;
    return d;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    if (argc < 3) { printf("usage: r5_probe <libxfeat_hip.so> <weights.bin> [scan launches per position]\n"); return 1; }
    lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { printf("dlopen: %s\n", dlerror()); return 2; }
    auto xfh_create = sym<int (*)(const float* const*, int, int, H*)>("xfh_create");
    auto xfh_last_error = sym<const char* (*)()>("xfh_last_error");
    auto xfh_set_option = sym<int (*)(H, const char*, int)>("xfh_set_option");
    auto xfh_set_status_buffer = sym<int (*)(H, int32_t*)>("xfh_set_status_buffer");
    auto xfh_debug_block1 = reinterpret_cast<int (*)(H, const float*, const float*, int, int, int, float*, void*)>(dlsym(lib, "xfh_debug_block1"));
    auto xfh_debug_head_soak = sym<int (*)(H, const float*, int, int, int, int, float*, float*, double*, float*, const float*, float*, const float*, int, int, int, unsigned*, unsigned*, unsigned, void*)>("xfh_debug_head_soak");
    auto xfh_backbone_workspace_bytes = sym<size_t (*)(int, int, int, int)>("xfh_backbone_workspace_bytes");
    auto xfh_backbone = sym<int (*)(H, const float*, int, int, int, int, float*, float*, float*, float*, float*, void*, size_t, void*)>("xfh_backbone");
    // ---- weights
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
    int32_t* status;
    HIPCHK(hipMalloc(&status, 4)); HIPCHK(hipMemset(status, 0, 4));
    xfh_set_status_buffer(h, status);
    auto take_status = [&] { int32_t v; HIPCHK(hipMemcpy(&v, status, 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemset(status, 0, 4)); return v; };
    const int B = 64, Hh = 480, W = 640;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    auto timed = [&](int n, auto fn) { for (int i = 0; i < 3; ++i) fn(); HIPCHK(hipDeviceSynchronize()); HIPCHK(hipEventRecord(e0, 0)); for (int i = 0; i < n; ++i) fn(); HIPCHK(hipEventRecord(e1, 0));
                                       HIPCHK(hipEventSynchronize(e1)); float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1)); return 1e3 * ms / n; };
    // ---- inputs
    const size_t npx = (size_t)B * Hh * W;
    auto hgray = rnd(npx, 1, 0.f, 1.f);
    std::vector<float> hcoef(2 * B);
    for (int b = 0; b < B; ++b) { hcoef[2 * b] = 3.4f + 0.01f * b; hcoef[2 * b + 1] = -1.7f; }
    float *gray, *coef;
    HIPCHK(hipMalloc(&gray, npx * 4)); HIPCHK(hipMalloc(&coef, 2 * B * 4));
    HIPCHK(hipMemcpy(gray, hgray.data(), npx * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(coef, hcoef.data(), 2 * B * 4, hipMemcpyHostToDevice));
    const bool scan = argc > 3;
    if (!scan) {
        // ---- block1
        if (xfh_debug_block1) {
            const size_t nx1 = (size_t)B * 24 * (Hh / 4) * (W / 4);
            float* x1; HIPCHK(hipMalloc(&x1, nx1 * 4));
            std::vector<float> ref(nx1), got(nx1);
            for (int mode : {5, 6, 7, 5}) {
                if (xfh_set_option(h, "block1", mode)) { printf("block1 = %d: %s\n", mode, xfh_last_error()); continue; }
                HIPCHK(hipMemset(x1, 0xff, nx1 * 4));
                if (xfh_debug_block1(h, gray, coef, B, Hh, W, x1, nullptr)) { printf("block1 %d: %s\n", mode, xfh_last_error()); continue; }
                const double us = timed(20, [&] { xfh_debug_block1(h, gray, coef, B, Hh, W, x1, nullptr); });
                HIPCHK(hipMemcpy(got.data(), x1, nx1 * 4, hipMemcpyDeviceToHost));
                if (mode == 5) ref = got;
                double am; size_t nb;
                const double d = maxdiff(got, ref, &am, &nb, 1e-4);
                printf("block1 mode %d: %8.1f us per launch; vs mode 5: max |diff| %.3g (max |x1| %.3g), %zu values beyond 1e-4; status %d\n", mode, us, d, am, nb, take_status());
            }
            xfh_set_option(h, "block1", 0);
            HIPCHK(hipFree(x1));
        } else printf("(no xfh_debug_block1 in this build)\n");
        // ---- heads
        {
            float *heat, *coefd; double* part;
            HIPCHK(hipMalloc(&heat, npx * 4)); HIPCHK(hipMalloc(&part, 8 * B * 128)); coefd = coef;
            std::vector<float> ref(npx), got(npx);
            for (int v : {101, 100, 0, 102, 103, 104, 101}) {
                HIPCHK(hipMemset(heat, 0xff, npx * 4));
                if (xfh_debug_head_soak(h, nullptr, B, 3, Hh, W, gray, coefd, part, heat, nullptr, nullptr, nullptr, v, 1, 0, nullptr, nullptr, 0, nullptr)) { printf("head variant %d: %s\n", v, xfh_last_error()); continue; }
                HIPCHK(hipMemcpy(got.data(), heat, npx * 4, hipMemcpyDeviceToHost));
                const double us = timed(4, [&] { xfh_debug_head_soak(h, nullptr, B, 3, Hh, W, gray, coefd, part, heat, nullptr, nullptr, nullptr, v, 10, 0, nullptr, nullptr, 0, nullptr); }) / 10;
                if (v == 101 && ref[0] == 0.f && ref[1] == 0.f) ref = got;
                double am; size_t nb;
                const double d = maxdiff(got, ref, &am, &nb, 1e-5);
                printf("key-point head variant %3d: %8.1f us per launch; heat vs variant 101: max |diff| %.3g (max %.3g), %zu values beyond 1e-5; status %d\n", v, us, d, am, nb, take_status());
            }
            HIPCHK(hipFree(heat)); HIPCHK(hipFree(part));
        }
        // ---- backbone under option sets
        {
            auto himg = rnd(3 * npx, 7, 0.f, 1.f);
            float* img; HIPCHK(hipMalloc(&img, 3 * npx * 4)); HIPCHK(hipMemcpy(img, himg.data(), 3 * npx * 4, hipMemcpyHostToDevice));
            const size_t ncell = (size_t)B * (Hh / 8) * (W / 8), wsb = xfh_backbone_workspace_bytes(B, 3, Hh, W);
            float *feats, *heat, *rel; void* ws;
            HIPCHK(hipMalloc(&feats, ncell * 64 * 4)); HIPCHK(hipMalloc(&heat, npx * 4)); HIPCHK(hipMalloc(&rel, ncell * 4)); HIPCHK(hipMalloc(&ws, wsb + 256));
            void* wsa = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~uintptr_t(255));
            struct Cfg { const char* name; int block1, heads, fx; };
            const Cfg cfgs[] = {{"default (heads_f32 2 = dustbin dot)", 0, 2, 3}, {"heads_f32 3 (round-4 heads)", 0, 3, 3}, {"block1 6", 6, 2, 3}, {"block1 7", 7, 2, 3}, {"fx 7 (two-fragment conv_bx64)", 0, 2, 7},
                                {"fx 67 (split-format link)", 0, 2, 67}, {"bf16 heads", 0, 0, 3}, {"fx heads (fx 11)", 0, 0, 11}, {"fx heads, two fragments (fx 27)", 0, 0, 27},
                                {"fx heads, B through LDS (fx 43)", 0, 0, 43}, {"block1 7 + fx 67 + fx heads", 7, 0, 75}, {"default again", 0, 2, 3}};
            std::vector<float> rf(ncell * 64), rh(npx), rr(ncell), gf(ncell * 64), gh(npx), gr(ncell);
            bool have = false;
            for (const Cfg& c : cfgs) {
                if (xfh_set_option(h, "block1", c.block1) || xfh_set_option(h, "heads_f32", c.heads) || xfh_set_option(h, "fx", c.fx)) { printf("%s: %s\n", c.name, xfh_last_error()); continue; }
                auto run = [&] { return xfh_backbone(h, img, B, 3, Hh, W, feats, nullptr, heat, rel, nullptr, wsa, wsb, nullptr); };
                if (run()) { printf("%s: %s\n", c.name, xfh_last_error()); continue; }
                const double us = timed(8, [&] { run(); });
                HIPCHK(hipMemcpy(gf.data(), feats, gf.size() * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(gh.data(), heat, gh.size() * 4, hipMemcpyDeviceToHost));
                HIPCHK(hipMemcpy(gr.data(), rel, gr.size() * 4, hipMemcpyDeviceToHost));
                if (!have) { rf = gf; rh = gh; rr = gr; have = true; }
                double af, ah;
                const double df = maxdiff(gf, rf, &af), dh = maxdiff(gh, rh, &ah), dr = maxdiff(gr, rr);
                printf("backbone %-40s %8.1f us; vs default: feats %.3g (max %.3g), heat %.3g (max %.3g), reliability %.3g; status %d\n", c.name, us, df, af, dh, ah, dr, take_status());
            }
        }
    } else {
        // ---- cold-start position scan of the split heads (needs the --scan build)
        const int n = atoi(argv[3]);
        float *heat, *href; double* part; unsigned* rep;
        HIPCHK(hipMalloc(&heat, npx * 4)); HIPCHK(hipMalloc(&href, npx * 4)); HIPCHK(hipMalloc(&part, 8 * B * 128)); HIPCHK(hipMalloc(&rep, (4 + 4 * 1024) * 4));
        const int base[3] = {0, 102, 104}, cold[3] = {1000, 4000, 5000};
        const char* nm[3] = {"bf16 head (the box's control)", "fp16-pair head", "fp16-pair head, B through LDS"};
        const int kinds = argc > 4 ? atoi(argv[4]) : 7;      // bit 0: bf16 head, 1: fp16 pair, 2: fp16 pair with B through LDS
        for (int k = 0; k < 3; ++k) {
            if (!((kinds >> k) & 1)) continue;
            if (xfh_debug_head_soak(h, nullptr, B, 3, Hh, W, gray, coef, part, href, nullptr, nullptr, nullptr, base[k], 1, 0, nullptr, nullptr, 0, nullptr)) { printf("variant %d: %s\n", base[k], xfh_last_error()); continue; }
            HIPCHK(hipDeviceSynchronize());
            std::string line;
            long total = 0;
            for (int s = 0; s < 16; ++s) {
                HIPCHK(hipMemset(rep, 0, (4 + 4 * 1024) * 4));
                if (xfh_debug_head_soak(h, nullptr, B, 3, Hh, W, gray, coef, part, heat, href, nullptr, nullptr, cold[k] + s, n, 0, rep, nullptr, 1024, nullptr)) { line += " n/a"; continue; }
                HIPCHK(hipDeviceSynchronize());
                std::vector<unsigned> r(4 + 4 * 1024);
                HIPCHK(hipMemcpy(r.data(), rep, r.size() * 4, hipMemcpyDeviceToHost));
                // distinct launches among the recorded float4s
                std::vector<unsigned> its;
                for (unsigned i = 0; i < std::min(r[0], 1024u); ++i) its.push_back(r[4 + 4 * i]);
                std::sort(its.begin(), its.end()); its.erase(std::unique(its.begin(), its.end()), its.end());
                line += " " + std::to_string(its.size()) + (r[0] > 1024 ? "+" : "");
                total += (long)its.size();
            }
            printf("%-34s cold-started, %d launches at each of 16 code positions: launches with a wrong heat map:%s   (total %ld)\n", nm[k], n, line.c_str(), total);
        }
    }
    if (scan && xfh_debug_block1) {
        // ---- block1 cold-started (xfh_debug_cold_start: every workgroup of a matrix-core kernel begins on an invalidated instruction cache): modes 6 / 7 against their quiet result
        auto xfh_debug_cold_start = sym<int (*)(int)>("xfh_debug_cold_start");
        const int Bs = 8;
        const size_t nx1 = (size_t)Bs * 24 * (Hh / 4) * (W / 4);
        float* x1; HIPCHK(hipMalloc(&x1, nx1 * 4));
        std::vector<float> ref(nx1), got(nx1);
        const int nl = std::max(200, atoi(argv[3]) / 8);
        for (int mode : {6, 7}) {
            if (xfh_set_option(h, "block1", mode)) continue;
            xfh_debug_cold_start(0);
            xfh_debug_block1(h, gray, coef, Bs, Hh, W, x1, nullptr);
            HIPCHK(hipMemcpy(ref.data(), x1, nx1 * 4, hipMemcpyDeviceToHost));
            xfh_debug_cold_start(1);
            int wrong = 0;
            for (int i = 0; i < nl; ++i) {
                xfh_debug_block1(h, gray, coef, Bs, Hh, W, x1, nullptr);
                HIPCHK(hipMemcpy(got.data(), x1, nx1 * 4, hipMemcpyDeviceToHost));
                wrong += memcmp(got.data(), ref.data(), nx1 * 4) != 0;
            }
            xfh_debug_cold_start(0);
            printf("block1 mode %d cold-started, B = %d: %d of %d launches differ from the quiet result; status %d\n", mode, Bs, wrong, nl, take_status());
        }
        xfh_set_option(h, "block1", 0);
        // ---- the whole backbone, B = 8, cold-started, with the prepared forms on (block1 7, fp16-pair heads, two-fragment conv_bx64): every output against the quiet run
        {
            const int Bb = 8;
            const size_t np8 = (size_t)Bb * Hh * W, nc8 = (size_t)Bb * (Hh / 8) * (W / 8), wsb = xfh_backbone_workspace_bytes(Bb, 3, Hh, W);
            auto himg = rnd(3 * np8, 11, 0.f, 1.f);
            float *img, *feats, *heat, *rel; void* ws;
            HIPCHK(hipMalloc(&img, 3 * np8 * 4)); HIPCHK(hipMemcpy(img, himg.data(), 3 * np8 * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMalloc(&feats, nc8 * 64 * 4)); HIPCHK(hipMalloc(&heat, np8 * 4)); HIPCHK(hipMalloc(&rel, nc8 * 4)); HIPCHK(hipMalloc(&ws, wsb + 256));
            void* wsa = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~uintptr_t(255));
            struct Cfg { const char* name; int block1, heads, fx; };
            for (const Cfg& c : {Cfg{"block1 7 + fx heads + fx 7", 7, 0, 15}, Cfg{"defaults of the branch", 0, 2, 3}}) {
                xfh_set_option(h, "block1", c.block1); xfh_set_option(h, "heads_f32", c.heads); xfh_set_option(h, "fx", c.fx);
                std::vector<float> rf(nc8 * 64), rh(np8), gf(nc8 * 64), gh(np8);
                xfh_debug_cold_start(0);
                if (xfh_backbone(h, img, Bb, 3, Hh, W, feats, nullptr, heat, rel, nullptr, wsa, wsb, nullptr)) { printf("%s: %s\n", c.name, xfh_last_error()); continue; }
                HIPCHK(hipMemcpy(rf.data(), feats, rf.size() * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(rh.data(), heat, rh.size() * 4, hipMemcpyDeviceToHost));
                xfh_debug_cold_start(1);
                int wrong = 0;
                const int nb = std::max(100, atoi(argv[3]) / 40);
                for (int i = 0; i < nb; ++i) {
                    xfh_backbone(h, img, Bb, 3, Hh, W, feats, nullptr, heat, rel, nullptr, wsa, wsb, nullptr);
                    HIPCHK(hipMemcpy(gf.data(), feats, gf.size() * 4, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(gh.data(), heat, gh.size() * 4, hipMemcpyDeviceToHost));
                    wrong += memcmp(gf.data(), rf.data(), gf.size() * 4) != 0 || memcmp(gh.data(), rh.data(), gh.size() * 4) != 0;
                }
                xfh_debug_cold_start(0);
                printf("backbone cold-started, B = %d, %s: %d of %d runs differ from the quiet one (feats or heat); status %d\n", Bb, c.name, wrong, nb, take_status());
            }
        }
    }
    printf("done\n");
    return 0;
}
