// Stand-alone probe of conv_rs64_kernel (conv_rs64_body.hpp: 64 -> 64 3x3 in the fp16-pair arithmetic with the weights resident in registers) through the C ABI -- no Python,
// no torch: a gpurun call with a prebuilt binary costs ~15 s of GPU budget.  For block4.1 (VGA 1/16 scale: 30 x 40), block_fusion.0 (1/8: 60 x 80) and a ragged map, batch 64:
//   xfh_conv_layer variant 1 (generic direct kernel: the reference), 11 (conv_bx64_kernel, fp16 pair), 12 (conv_rs64_kernel): max |diff| against variant 1, HIP-event time per
//   launch; then `repeats` cold-started launches (xfh_debug_cold_start) of variant 12 compared bit for bit with its first result; then the backbone with fx = 3 / 131 / 387 (387: all five 64 -> 64 launches on conv_rs64_kernel).
//     hipcc -O2 -w tools/bench_src/rs64_probe.cpp -o gpurun_probe/rs64_probe -ldl ; gpurun_probe/rs64_probe <libxfeat_hip.so> <weights.bin> [repeats = 200]
//     (weights.bin: int32 count, then per array int32 n + n floats -- the arrays of XFeatModel.weight_arrays(); tools/ab_configs.py --dump-weights writes it)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

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

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    if (argc < 3) { printf("usage: rs64_probe <libxfeat_hip.so> <weights.bin> [cold repeats]\n"); return 1; }
    const int repeats = argc > 3 ? atoi(argv[3]) : 200;
    lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { printf("dlopen: %s\n", dlerror()); return 2; }
    auto xfh_create = sym<int (*)(const float* const*, int, int, H*)>("xfh_create");
    auto xfh_last_error = sym<const char* (*)()>("xfh_last_error");
    auto xfh_set_option = sym<int (*)(H, const char*, int)>("xfh_set_option");
    auto xfh_set_status_buffer = sym<int (*)(H, int32_t*)>("xfh_set_status_buffer");
    auto xfh_conv_layer = sym<int (*)(H, int, const float*, int, int, int, float*, int, void*)>("xfh_conv_layer");
    auto xfh_debug_cold_start = sym<int (*)(int)>("xfh_debug_cold_start");
    auto xfh_backbone_workspace_bytes = sym<size_t (*)(int, int, int, int)>("xfh_backbone_workspace_bytes");
    auto xfh_backbone = sym<int (*)(H, const float*, int, int, int, int, float*, float*, float*, float*, float*, void*, size_t, void*)>("xfh_backbone");
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
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    auto timed = [&](int n, auto fn) { for (int i = 0; i < 3; ++i) fn(); HIPCHK(hipDeviceSynchronize()); HIPCHK(hipEventRecord(e0, 0)); for (int i = 0; i < n; ++i) fn(); HIPCHK(hipEventRecord(e1, 0));
                                       HIPCHK(hipEventSynchronize(e1)); float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1)); return 1e3 * ms / n; };
    // layer indices of accelerated_features_amd/spec.py: CONVS (skip1.1 = 0, block1.0-.3 = 1-4, block2.0-.1, block3.0-.2, block4.0-.2 = 10-12, block5.0-.3, block_fusion.0-.2 = 17-19, heads)
    struct Case { const char* name; int layer, B, Hm, Wm; };
    const Case cases[] = {{"block4.1  B 64  30 x 40", 11, 64, 30, 40}, {"block4.2  B 64  30 x 40", 12, 64, 30, 40}, {"block_fusion.0  B 64  60 x 80", 17, 64, 60, 80},
                          {"block_fusion.0  B 3  41 x 93", 17, 3, 41, 93}, {"block4.1  B 1  30 x 40", 11, 1, 30, 40}, {"block5.1  B 64  15 x 20 (128 ch)", 14, 64, 15, 20}, {"block4.0  B 64  60 x 80 (stride 2: variants 1 / 10 / 11)", 10, 64, 60, 80}, {"block5.0  B 64  30 x 40 (stride 2)", 13, 64, 30, 40},
                          {"block_fusion.0  B 8  128 x 128 (2 strips)", 17, 8, 128, 128}, {"block_fusion.0  B 2  150 x 200 (2 strips)", 17, 2, 150, 200}, {"block5.1  B 8  32 x 100 (128 ch, 2 strips)", 14, 8, 32, 100}};
    if (argc > 3 && !strcmp(argv[3], "one")) {      // one layer, one variant, a few launches: the command rocprofv3 --pmc runs (argv: lib weights one <case> <variant> [launches])
        // (case >= 100: the 24-channel layers at 1/4 scale, VGA batch 64 -- 100 = block2.0, 101 = block2.1 (24 -> 24), 102 = block3.0 (24 -> 64, stride 2): the default kernel is variant 0)
        static const Case c24[] = {{"block2.0  B 64  120 x 160 (24 ch)", 5, 64, 120, 160}, {"block2.1  B 64  120 x 160 (24 ch)", 6, 64, 120, 160}, {"block3.0  B 64  120 x 160 (24 -> 64, stride 2)", 7, 64, 120, 160}};
        const int ci = atoi(argv[4]);
        const Case& c = ci >= 100 ? c24[ci - 100] : cases[ci];
        const int v = atoi(argv[5]), nl = argc > 6 ? atoi(argv[6]) : 5;
        const int nch = ci >= 100 ? 24 : c.layer == 14 || c.layer == 15 ? 128 : 64;
        const size_t n = (size_t)c.B * 128 * c.Hm * c.Wm;      // (room for the widest output of any case)
        auto hx = rnd((size_t)c.B * nch * c.Hm * c.Wm, (unsigned)c.layer + c.B, -1.f, 3.f);
        float *x, *y;
        HIPCHK(hipMalloc(&x, hx.size() * 4)); HIPCHK(hipMalloc(&y, n * 4));
        HIPCHK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        for (int i = 0; i < nl; ++i) if (xfh_conv_layer(h, c.layer, x, c.B, c.Hm, c.Wm, y, v, nullptr)) { printf("%s\n", xfh_last_error()); return 2; }
        HIPCHK(hipDeviceSynchronize());
        printf("%s variant %d: %d launches\n", c.name, v, nl);
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "trace24")) {      // conv_bx_kernel<24,24,fx>'s own stamps (k_conv_bx.hip: BX_STAMP 0 tile start, 1 staged, 2 barrier passed, 3 MFMAs issued; 6 per tile, 10 tiles, 64 slots per workgroup)
        auto xfh_debug_trace = reinterpret_cast<int (*)(H, long long*)>(dlsym(lib, "xfh_debug_trace"));
        const int B = 64, Hm = 120, Wm = 160;
        const size_t n = (size_t)B * 24 * Hm * Wm;
        auto hx = rnd(n, 5, 0.f, 3.f);
        float *x, *y; long long* tr;
        const size_t ntr = ((size_t)1 << 21) + ((size_t)1 << 17);
        HIPCHK(hipMalloc(&x, n * 4)); HIPCHK(hipMalloc(&y, n * 4)); HIPCHK(hipMalloc(&tr, ntr * 8));
        HIPCHK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
        for (int i = 0; i < 3; ++i) xfh_conv_layer(h, 5, x, B, Hm, Wm, y, 0, nullptr);
        HIPCHK(hipMemset(tr, 0, ntr * 8));
        xfh_debug_trace(h, tr);
        xfh_conv_layer(h, 5, x, B, Hm, Wm, y, 0, nullptr);
        HIPCHK(hipDeviceSynchronize());
        xfh_debug_trace(h, nullptr);
        std::vector<long long> t(512 * 64);
        HIPCHK(hipMemcpy(t.data(), tr, t.size() * 8, hipMemcpyDeviceToHost));
        double ph[4] = {0, 0, 0, 0}, per = 0; long cnt = 0, cntp = 0;
        double by_alloc[2][5] = {{0}}; long na[2] = {0, 0};
        for (int g = 0; g < 512; ++g) {
            const long long* q = &t[g * 64];
            const int second = (q[62] & 0xff) != 0;
            for (int k = 1; k < 8; ++k) {      // tiles 1 .. 7 of the workgroup (steady state)
                const long long* a = q + k * 6;
                if (!a[0] || !a[3] || !a[6]) continue;
                ph[0] += (double)(a[1] - a[0]); ph[1] += (double)(a[2] - a[1]); ph[2] += (double)(a[3] - a[2]); ph[3] += (double)(a[6] - a[3]); ++cnt;
                per += (double)(a[6] - a[0]); ++cntp;
                by_alloc[second][0] += (double)(a[1] - a[0]); by_alloc[second][1] += (double)(a[2] - a[1]); by_alloc[second][2] += (double)(a[3] - a[2]); by_alloc[second][3] += (double)(a[6] - a[3]); ++na[second];
            }
        }
        printf("conv_bx_kernel<24,24,fx> block2.0 B 64: per tile (wave 0 of %ld workgroup-tiles, cycles): stage (wait for the tile's loads + convert + LDS writes) %.0f, barrier %.0f, MFMAs (84) + next tile's loads issued %.0f, epilogue (24 stores) -> next tile %.0f; period %.0f\n",
               cnt, ph[0] / cnt, ph[1] / cnt, ph[2] / cnt, ph[3] / cnt, per / cntp);
        for (int s2 = 0; s2 < 2; ++s2) if (na[s2]) printf("  %s workgroup of a CU: stage %.0f barrier %.0f MFMA %.0f epilogue %.0f\n", s2 ? "second" : "first", by_alloc[s2][0] / na[s2], by_alloc[s2][1] / na[s2], by_alloc[s2][2] / na[s2], by_alloc[s2][3] / na[s2]);
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "lag")) {      // experiment: conv_bx_kernel<24,24,fx> under different phase lags of a CU's second workgroup (XFH_DBG_LAG, experiment builds only)
        const int B = 64, Hm = 120, Wm = 160;
        const size_t n = (size_t)B * 24 * Hm * Wm;
        auto hx = rnd(n, 5, 0.f, 3.f);
        float *x, *y;
        HIPCHK(hipMalloc(&x, n * 4)); HIPCHK(hipMalloc(&y, (size_t)B * 64 * Hm * Wm * 4));
        HIPCHK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
        for (int layer : {5, 7}) {
            for (int lag : {11, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 16, 20, 24, 11}) {
                char buf[16]; snprintf(buf, sizeof buf, "%d", lag); setenv("XFH_DBG_LAG", buf, 1);
                if (xfh_conv_layer(h, layer, x, B, Hm, Wm, y, 0, nullptr)) { printf("%s\n", xfh_last_error()); return 2; }
                const double us = timed(30, [&] { xfh_conv_layer(h, layer, x, B, Hm, Wm, y, 0, nullptr); });
                printf("layer %d lag %2d: %7.1f us\n", layer, lag, us);
            }
        }
        return 0;
    }
    for (const Case& c : cases) {
        const int nch = c.layer == 14 || c.layer == 15 ? 128 : 64;
        const bool s2 = c.layer == 10 || c.layer == 13;      // stride 2: output 64 | 128 channels at half the size
        const size_t n = (size_t)c.B * nch * c.Hm * c.Wm, nout = s2 ? (size_t)c.B * (c.layer == 13 ? 128 : 64) * (c.Hm / 2) * (c.Wm / 2) : n;
        auto hx = rnd(n, (unsigned)c.layer + c.B, -1.f, 3.f);
        float *x, *y;
        HIPCHK(hipMalloc(&x, n * 4)); HIPCHK(hipMalloc(&y, nout * 4));
        HIPCHK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
        std::vector<float> ref(nout), got(nout), first(nout);
        for (int v : {1, 10, 11, 12}) {
            if ((v == 11 && nch == 128) || (v == 12 && s2) || (v == 10 && !s2)) continue;      // (no conv_bx64 form of the 128-channel layers: the backbone runs them as Winograd)
            HIPCHK(hipMemset(y, 0xff, nout * 4));
            if (xfh_conv_layer(h, c.layer, x, c.B, c.Hm, c.Wm, y, v, nullptr)) { printf("%s variant %d: %s\n", c.name, v, xfh_last_error()); continue; }
            HIPCHK(hipMemcpy(got.data(), y, nout * 4, hipMemcpyDeviceToHost));
            const double us = timed(20, [&] { xfh_conv_layer(h, c.layer, x, c.B, c.Hm, c.Wm, y, v, nullptr); });
            if (v == 1) ref = got;
            if (v == (s2 ? 11 : 12)) first = got;
            double d = 0, m = 0;
            for (size_t i = 0; i < nout; ++i) { const double e = std::fabs((double)got[i] - ref[i]); if (!(e <= d)) d = e; m = std::fmax(m, std::fabs((double)ref[i])); }
            printf("%-32s variant %2d: %8.1f us per launch; vs variant 1: max |diff| %.3g (max |y| %.3g); status %d\n", c.name, v, us, d, m, take_status());
        }
        // cold instruction cache: every launch must reproduce the first result bit for bit
        xfh_debug_cold_start(1);
        size_t bad = 0;
        for (int r = 0; r < repeats; ++r) {
            HIPCHK(hipMemset(y, 0xff, nout * 4));
            if (xfh_conv_layer(h, c.layer, x, c.B, c.Hm, c.Wm, y, s2 ? 11 : 12, nullptr)) break;
            HIPCHK(hipMemcpy(got.data(), y, nout * 4, hipMemcpyDeviceToHost));
            if (memcmp(got.data(), first.data(), nout * 4)) ++bad;
        }
        xfh_debug_cold_start(0);
        printf("%-32s variant %d, %d cold-started launches: %zu differ from the first\n", c.name, s2 ? 11 : 12, repeats, bad);
        HIPCHK(hipFree(x)); HIPCHK(hipFree(y));
    }
    // ---- the 3x3 + 1x1 pairs as one launch: conv_rs64_kernel (13 NCHW / 14 channels-last) against conv_bx64_kernel (15 / 16); parity of 13 vs 15 and 14 vs 16
    for (const Case& c : {Case{"block3.1 + .2  B 64  60 x 80", 8, 64, 60, 80}, Case{"block_fusion.1 + .2  B 64  60 x 80", 18, 64, 60, 80}, Case{"block_fusion.1 + .2  B 3  41 x 61", 18, 3, 41, 61}, Case{"block_fusion.1 + .2  B 8  128 x 128 (2 strips)", 18, 8, 128, 128}, Case{"block3.1 + .2  B 2  150 x 200 (3 strips)", 8, 2, 150, 200}}) {
        const size_t n = (size_t)c.B * 64 * c.Hm * c.Wm;
        auto hx = rnd(n, (unsigned)c.layer + c.B, -1.f, 3.f);
        float *x, *y;
        HIPCHK(hipMalloc(&x, n * 4)); HIPCHK(hipMalloc(&y, n * 4));
        HIPCHK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
        std::vector<float> ref[2], got(n);
        for (int v : {15, 16, 13, 14}) {
            HIPCHK(hipMemset(y, 0xff, n * 4));
            if (xfh_conv_layer(h, c.layer, x, c.B, c.Hm, c.Wm, y, v, nullptr)) { printf("%s variant %d: %s\n", c.name, v, xfh_last_error()); continue; }
            HIPCHK(hipMemcpy(got.data(), y, n * 4, hipMemcpyDeviceToHost));
            const double us = timed(20, [&] { xfh_conv_layer(h, c.layer, x, c.B, c.Hm, c.Wm, y, v, nullptr); });
            if (v >= 15) ref[v - 15] = got;
            const std::vector<float>& r = ref[(v - 13) & 1];
            double d = 0, m = 0;
            if (r.size() == n) for (size_t i = 0; i < n; ++i) { const double e = std::fabs((double)got[i] - r[i]); if (!(e <= d)) d = e; m = std::fmax(m, std::fabs((double)r[i])); }
            printf("%-36s variant %2d: %8.1f us per launch; vs variant %d: max |diff| %.3g (max |y| %.3g); status %d\n", c.name, v, us, 15 + ((v - 13) & 1), d, m, take_status());
        }
        HIPCHK(hipFree(x)); HIPCHK(hipFree(y));
    }
    // ---- where a unit's cycles go: the stamped twin of the kernel (xfh_debug_trace; conv_rs64_body.hpp documents the 15 stamps per workgroup)
    if (void* pt = dlsym(lib, "xfh_debug_trace")) {
        auto xfh_debug_trace = reinterpret_cast<int (*)(H, long long*)>(pt);
        const size_t ntr = ((size_t)1 << 21) + ((size_t)1 << 17);
        long long* tr; HIPCHK(hipMalloc(&tr, ntr * 8));
        for (const Case& c : {cases[0], cases[2]}) {
            const size_t n = (size_t)c.B * 64 * c.Hm * c.Wm;
            auto hx = rnd(n, 99, -1.f, 3.f);
            float *x, *y;
            HIPCHK(hipMalloc(&x, n * 4)); HIPCHK(hipMalloc(&y, n * 4));
            HIPCHK(hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice));
            for (int i = 0; i < 3; ++i) xfh_conv_layer(h, c.layer, x, c.B, c.Hm, c.Wm, y, 12, nullptr);
            HIPCHK(hipMemset(tr, 0, ntr * 8));
            xfh_debug_trace(h, tr);
            xfh_conv_layer(h, c.layer, x, c.B, c.Hm, c.Wm, y, 12, nullptr);
            HIPCHK(hipDeviceSynchronize());
            xfh_debug_trace(h, nullptr);
            std::vector<long long> t(256 * 128);
            HIPCHK(hipMemcpy(t.data(), tr, t.size() * 8, hipMemcpyDeviceToHost));
            double sum[16] = {0}; int nw = 0;
            double slot[54] = {0};
            const char* what[13] = {"entry -> weights in registers", "-> first ring filled", "", "unit: taps 0-2 issued", "taps 3-4 issued (at the barrier)", "barrier passed", "taps 5-8 + reduction issued (first block done)",
                                    "second block: taps 0-2 + conversion", "taps 3-4 (at the barrier)", "barrier passed", "taps 5-8 + reduction", "segment stored (unit end)", ""};
            for (int g = 0; g < 256; ++g) {
                const long long* q = &t[g * 128];
                if (!q[0] || !q[13] || q[14] < 2 || !q[12]) continue;
                ++nw;
                sum[0] += (double)(q[1] - q[0]); sum[1] += (double)(q[2] - q[1]);
                for (int k = 3; k < 12; ++k) sum[k] += (double)(q[k + 1] - q[k]);
                for (int k = 0; k < 54; ++k) slot[k] += (double)((k < 53 ? q[33 + k] : q[12]) - q[32 + k]);
                sum[12] += (double)(q[13] - q[0]); sum[13] += (double)q[14]; sum[14] += (double)(q[12] - q[3]);
            }
            printf("%s: %d workgroups with a second unit; s_memtime ticks (means): whole kernel %.0f for %.1f units; ONE unit %.0f (108 MFMAs: floor 3456 cycles)\n", c.name, nw, sum[12] / (nw ? nw : 1), sum[13] / (nw ? nw : 1), sum[14] / (nw ? nw : 1));
            for (int k = 0; k < 12; ++k) if (what[k][0]) printf("    %-52s %8.0f\n", what[k], sum[k] / (nw ? nw : 1));
            printf("    slot -> next slot (two MFMAs = 64 cycles + what the slot issues), first block:");
            for (int k = 0; k < 54; ++k) { if (k == 27) printf("\n    second block:"); printf(" %.0f", slot[k] / (nw ? nw : 1)); }
            printf("\n");
            HIPCHK(hipFree(x)); HIPCHK(hipFree(y));
        }
        HIPCHK(hipFree(tr));
    }
    // ---- the backbone with the three unfused 64 -> 64 layers on conv_rs64_kernel (fx bit 128)
    {
        const int B = 64, Hh = 480, W = 640;
        const size_t npx = (size_t)B * Hh * W;
        auto himg = rnd(3 * npx, 7, 0.f, 1.f);
        float* img; HIPCHK(hipMalloc(&img, 3 * npx * 4)); HIPCHK(hipMemcpy(img, himg.data(), 3 * npx * 4, hipMemcpyHostToDevice));
        const size_t ncell = (size_t)B * (Hh / 8) * (W / 8), wsb = xfh_backbone_workspace_bytes(B, 3, Hh, W);
        float *feats, *heat, *rel; void* ws;
        HIPCHK(hipMalloc(&feats, ncell * 64 * 4)); HIPCHK(hipMalloc(&heat, npx * 4)); HIPCHK(hipMalloc(&rel, ncell * 4)); HIPCHK(hipMalloc(&ws, wsb + 256));
        void* wsa = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~uintptr_t(255));
        std::vector<float> ref(ncell * 64), got(ncell * 64);
        for (int fx : {3, 131, 387, 899, 1027, 1923, 3}) {      // 131: the unfused 64 -> 64 layers on conv_rs64_kernel; 387: the 3x3 + 1x1 pairs too; 899: + block5.1 / block5.2 on the 128-channel form; 1027: the stride-2 layers in the fp16-pair arithmetic alone; 1923: all of it
            if (xfh_set_option(h, "fx", fx)) { printf("fx = %d: %s\n", fx, xfh_last_error()); continue; }
            if (xfh_backbone(h, img, B, 3, Hh, W, feats, nullptr, heat, rel, nullptr, wsa, wsb, nullptr)) { printf("backbone fx %d: %s\n", fx, xfh_last_error()); continue; }
            HIPCHK(hipMemcpy(got.data(), feats, ncell * 64 * 4, hipMemcpyDeviceToHost));
            const double us = timed(10, [&] { xfh_backbone(h, img, B, 3, Hh, W, feats, nullptr, heat, rel, nullptr, wsa, wsb, nullptr); });
            if (fx == 3 && ref[0] == 0.f && ref[1] == 0.f) ref = got;
            double d = 0, m = 0;
            for (size_t i = 0; i < got.size(); ++i) { const double e = std::fabs((double)got[i] - ref[i]); if (!(e <= d)) d = e; m = std::fmax(m, std::fabs((double)ref[i])); }
            printf("backbone fx %3d: %8.1f us; feats vs fx 3: max |diff| %.3g (max %.3g); status %d\n", fx, us, d, m, take_status());
        }
    }
    return 0;
}
