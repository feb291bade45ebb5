// Cold-start code-position scan of every default matrix-core kernel in ONE process (no Python: the whole scan is ~4 GPU-minutes).
//   build (CPU):  python -m accelerated_features_amd.build ; for n in 1..15: python -m accelerated_features_amd.build --shift n
//                 hipcc -O2 -w --offload-arch=gfx950 tools/bench_src/scan_probe.cpp -o gpurun_probe/scan_probe -ldl
//   run (GPU):    gpurun_probe/scan_probe accelerated_features_amd gpurun_probe/weights.bin <launches per kernel and position>
// libxfeat_hip.so (position 0) and libxfeat_hip_shift<N>.so (every matrix-core kernel moved by 4 N bytes), N = 1..15: each default kernel alone in a tight loop,
// every launch cold-started (xfh_debug_cold_start: each workgroup begins on an invalidated instruction cache), every result compared ON THE DEVICE with the quiet result.
// Rounds 4-5 ran the retired split-bf16 key-point head first as the box's positive control (a scan counted only on a box where that kernel failed:
// profiles/r05_scan_all_kernels.txt); round 6 deleted that kernel from the tree -- the last source that holds it is commit 9802e34.
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

__global__ void cmp_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, size_t n16, unsigned* __restrict__ flag) {
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 x = a[i], y = b[i];
        bad |= x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w;
    }
    if (bad) atomicOr(flag, 1u);
}
__global__ void tally_kernel(unsigned* flag, unsigned* total) { if (*flag) { *total += 1; *flag = 0; } }

static std::vector<float> rnd(size_t n, unsigned seed, float lo, float hi) {
    std::vector<float> v(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; v[i] = lo + (hi - lo) * ((s >> 8) * (1.0f / 16777216.0f)); }
    return v;
}

struct Lib {
    void* so = nullptr;
    H h = nullptr;
    int (*conv_layer)(H, int, const float*, int, int, int, float*, int, void*) = nullptr;
    int (*cold)(int) = nullptr;
    int (*block1)(H, const float*, const float*, int, int, int, float*, void*) = nullptr;
    int (*backbone)(H, const float*, int, int, int, int, float*, float*, float*, float*, float*, void*, size_t, void*) = nullptr;
    size_t (*backbone_ws)(int, int, int, int) = nullptr;
    int (*match)(H, const float*, size_t, const float*, size_t, const uint16_t*, const uint16_t*, const int32_t*, const int32_t*, int, int, int, int, int, float, int64_t*, int64_t*, int32_t*, void*, size_t, void*) = nullptr;
    size_t (*match_ws)(int, int, int) = nullptr;
    int (*fine)(H, const float*, int, float*, void*, size_t, void*) = nullptr;
    size_t (*refine_ws)(int, int) = nullptr;
    int (*refine)(H, const float*, const float*, const float*, const float*, const float*, const int64_t*, const int64_t*, const int32_t*, int, int, float, float*, int32_t*, void*, size_t, void*) = nullptr;
    const char* (*last_error)() = nullptr;
};
static bool open_lib(Lib& L, const std::string& path, const std::vector<const float*>& ptrs) {
    L.so = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!L.so) { printf("dlopen %s: %s\n", path.c_str(), dlerror()); return false; }
#define SYM(f, name) L.f = reinterpret_cast<decltype(L.f)>(dlsym(L.so, name)); if (!L.f) { printf("%s: missing %s\n", path.c_str(), name); return false; }
    SYM(conv_layer, "xfh_conv_layer") SYM(cold, "xfh_debug_cold_start") SYM(block1, "xfh_debug_block1") SYM(backbone, "xfh_backbone")
    SYM(backbone_ws, "xfh_backbone_workspace_bytes") SYM(match, "xfh_match_mnn") SYM(match_ws, "xfh_match_workspace_bytes") SYM(last_error, "xfh_last_error")
    SYM(fine, "xfh_fine_matcher") SYM(refine_ws, "xfh_refine_workspace_bytes") SYM(refine, "xfh_refine_matches")
    auto create = reinterpret_cast<int (*)(const float* const*, int, int, H*)>(dlsym(L.so, "xfh_create"));
    if (!create || create(ptrs.data(), (int)ptrs.size(), 0, &L.h)) { printf("%s: xfh_create failed: %s\n", path.c_str(), L.last_error ? L.last_error() : "?"); return false; }
    return true;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    if (argc < 4) { printf("usage: scan_probe <dir of the libraries> <weights.bin> <launches per kernel and position> [- fine]\n"); return 1; }
    const std::string dir = argv[1];
    const int N = atoi(argv[3]);
    FILE* f = fopen(argv[2], "rb");
    if (!f) { printf("cannot open %s\n", argv[2]); return 2; }
    int na = 0;
    if (fread(&na, 4, 1, f) != 1) return 2;
    std::vector<std::vector<float>> arrs(na);
    std::vector<const float*> ptrs(na);
    for (int i = 0; i < na; ++i) { int n; if (fread(&n, 4, 1, f) != 1) return 2; arrs[i].resize(n); if (fread(arrs[i].data(), 4, n, f) != (size_t)n) return 2; ptrs[i] = arrs[i].data(); }
    fclose(f);
    unsigned *flag, *total;
    HIPCHK(hipMalloc(&flag, 4)); HIPCHK(hipMalloc(&total, 4)); HIPCHK(hipMemset(flag, 0, 4));
    auto take_total = [&] { unsigned v; HIPCHK(hipMemcpy(&v, total, 4, hipMemcpyDeviceToHost)); return v; };

    // ---------------- 2. every default matrix-core kernel at 16 code positions (the whole library shifted), B = 16 VGA maps
    const int B = 16, Hh = 480, W = 640;
    struct K { const char* name; int layer, variant, div, cin, cout, stride; };
    // layer indices: accelerated_features_amd/spec.py (block2.0 = 5, block3.0 = 7, block3.1 = 8, block4.0 = 10, block4.1 = 11, block5.0 = 13, block5.1 = 14, block5.3 = 16, block_fusion.0 = 17, .1 = 18)
    const K ks[] = {{"conv_bx_kernel<24,24,fx> (block2.0)", 5, 0, 4, 24, 24, 1}, {"conv_bxs2_kernel<24,fx> (block3.0)", 7, 0, 4, 24, 64, 2},
                    {"conv_rs64_kernel<1> (block3.1 + .2)", 8, 4, 8, 64, 64, 1}, {"conv_bx64s2x_kernel<1> (block4.0)", 10, 0, 8, 64, 64, 2},
                    {"conv_rs64_kernel<0> (block4.1)", 11, 0, 16, 64, 64, 1}, {"conv_bx64s2x_kernel<2> (block5.0)", 13, 0, 16, 64, 128, 2},
                    {"conv_rs64_kernel<0,128> (block5.1)", 14, 0, 32, 128, 128, 1}, {"conv_rs64_kernel<0> (block_fusion.0)", 17, 0, 8, 64, 64, 1},
                    {"conv_rs64_kernel<2> (block_fusion.1 + .2)", 18, 5, 8, 64, 64, 1}};
    const bool only_fine = argc > 5 && !strcmp(argv[5], "fine");      // (a later visit for one kernel added after the round's scan: the control + linear_fx_kernel alone)
    const int nk = (int)(sizeof(ks) / sizeof(ks[0])), NEXTRA = 5;      // + block1_mx<7>, the fp16-pair heads (whole backbone), mnn_f16_sweep (+ refine)
    std::vector<std::vector<long>> wrong(nk + NEXTRA, std::vector<long>(16, -1));
    std::vector<long> runs(nk + NEXTRA, 0);
    for (int pos = 0; pos < 16; ++pos) {
        Lib L;
        if (!open_lib(L, dir + (pos ? "/libxfeat_hip_shift" + std::to_string(pos) + ".so" : "/libxfeat_hip.so"), ptrs)) { printf("position %d: library missing -- skipped\n", pos); continue; }
        for (int ki = 0; ki < (only_fine ? 0 : nk); ++ki) {
            const K& k = ks[ki];
            const int hin = Hh / k.div, win = W / k.div, ho = (hin - 1) / k.stride + 1, wo = (win - 1) / k.stride + 1;
            const size_t nin = (size_t)B * k.cin * hin * win, nout = (size_t)B * k.cout * ho * wo;
            auto hx = rnd(nin, 100 + ki, 0.f, 3.f);
            float *x, *y, *want;
            HIPCHK(hipMalloc(&x, nin * 4)); HIPCHK(hipMalloc(&y, nout * 4)); HIPCHK(hipMalloc(&want, nout * 4));
            HIPCHK(hipMemcpy(x, hx.data(), nin * 4, hipMemcpyHostToDevice));
            L.cold(0);
            if (L.conv_layer(L.h, k.layer, x, B, hin, win, want, k.variant, nullptr)) { printf("%s: %s\n", k.name, L.last_error()); HIPCHK(hipFree(x)); HIPCHK(hipFree(y)); HIPCHK(hipFree(want)); continue; }
            HIPCHK(hipDeviceSynchronize());
            HIPCHK(hipMemset(total, 0, 4));
            L.cold(1);
            for (int i = 0; i < N; ++i) {
                L.conv_layer(L.h, k.layer, x, B, hin, win, y, k.variant, nullptr);
                cmp_kernel<<<256, 256>>>(reinterpret_cast<const uint4*>(y), reinterpret_cast<const uint4*>(want), nout / 4, flag);
                tally_kernel<<<1, 1>>>(flag, total);
            }
            L.cold(0);
            HIPCHK(hipDeviceSynchronize());
            wrong[ki][pos] = take_total(); runs[ki] = N;
            HIPCHK(hipFree(x)); HIPCHK(hipFree(y)); HIPCHK(hipFree(want));
        }
        if (!only_fine) {   // block1_mx_kernel<7> alone
            const size_t npx = (size_t)B * Hh * W, nx1 = (size_t)B * 24 * (Hh / 4) * (W / 4);
            auto hgray = rnd(npx, 1, 0.f, 1.f);
            std::vector<float> hcoef(2 * B);
            for (int b = 0; b < B; ++b) { hcoef[2 * b] = 3.4f + 0.01f * b; hcoef[2 * b + 1] = -1.7f; }
            float *gray, *coef, *x1, *want;
            HIPCHK(hipMalloc(&gray, npx * 4)); HIPCHK(hipMalloc(&coef, 2 * B * 4)); HIPCHK(hipMalloc(&x1, nx1 * 4)); HIPCHK(hipMalloc(&want, nx1 * 4));
            HIPCHK(hipMemcpy(gray, hgray.data(), npx * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(coef, hcoef.data(), 2 * B * 4, hipMemcpyHostToDevice));
            L.cold(0);
            L.block1(L.h, gray, coef, B, Hh, W, want, nullptr);
            HIPCHK(hipDeviceSynchronize()); HIPCHK(hipMemset(total, 0, 4));
            L.cold(1);
            const int n1 = std::max(200, N / 4);
            for (int i = 0; i < n1; ++i) {
                L.block1(L.h, gray, coef, B, Hh, W, x1, nullptr);
                cmp_kernel<<<256, 256>>>(reinterpret_cast<const uint4*>(x1), reinterpret_cast<const uint4*>(want), nx1 / 4, flag);
                tally_kernel<<<1, 1>>>(flag, total);
            }
            L.cold(0);
            HIPCHK(hipDeviceSynchronize());
            wrong[nk][pos] = take_total(); runs[nk] = n1;
            HIPCHK(hipFree(gray)); HIPCHK(hipFree(coef)); HIPCHK(hipFree(x1)); HIPCHK(hipFree(want));
        }
        if (!only_fine) {   // the whole backbone (every kernel above once more in sequence + both fp16-pair heads): feats and heat against the quiet run, B = 8
            const int Bb = 8;
            const size_t np8 = (size_t)Bb * Hh * W, nc8 = (size_t)Bb * (Hh / 8) * (W / 8), wsb = L.backbone_ws(Bb, 3, Hh, W);
            auto himg = rnd(3 * np8, 11, 0.f, 1.f);
            float *img, *feats, *heat, *rel, *wf, *wh; void* ws;
            HIPCHK(hipMalloc(&img, 3 * np8 * 4)); HIPCHK(hipMemcpy(img, himg.data(), 3 * np8 * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMalloc(&feats, nc8 * 64 * 4)); HIPCHK(hipMalloc(&heat, np8 * 4)); HIPCHK(hipMalloc(&rel, nc8 * 4)); HIPCHK(hipMalloc(&ws, wsb + 256));
            HIPCHK(hipMalloc(&wf, nc8 * 64 * 4)); HIPCHK(hipMalloc(&wh, np8 * 4));
            void* wsa = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~uintptr_t(255));
            L.cold(0);
            if (L.backbone(L.h, img, Bb, 3, Hh, W, wf, nullptr, wh, rel, nullptr, wsa, wsb, nullptr)) printf("backbone: %s\n", L.last_error());
            HIPCHK(hipDeviceSynchronize()); HIPCHK(hipMemset(total, 0, 4));
            L.cold(1);
            const int nb = std::max(100, N / 10);
            for (int i = 0; i < nb; ++i) {
                L.backbone(L.h, img, Bb, 3, Hh, W, feats, nullptr, heat, rel, nullptr, wsa, wsb, nullptr);
                cmp_kernel<<<256, 256>>>(reinterpret_cast<const uint4*>(feats), reinterpret_cast<const uint4*>(wf), nc8 * 16, flag);
                cmp_kernel<<<256, 256>>>(reinterpret_cast<const uint4*>(heat), reinterpret_cast<const uint4*>(wh), np8 / 4, flag);
                tally_kernel<<<1, 1>>>(flag, total);
            }
            L.cold(0);
            HIPCHK(hipDeviceSynchronize());
            wrong[nk + 1][pos] = take_total(); runs[nk + 1] = nb;
            HIPCHK(hipFree(img)); HIPCHK(hipFree(feats)); HIPCHK(hipFree(heat)); HIPCHK(hipFree(rel)); HIPCHK(hipFree(ws)); HIPCHK(hipFree(wf)); HIPCHK(hipFree(wh));
        }
        if (!only_fine) {   // the matcher: mnn_f16_sweep_kernel (cold hook in the shifted builds only) + refine, 8 pairs of 2048 unit rows: match lists against the quiet run
            const int P = 8, Nk = 2048;
            auto hd = rnd((size_t)2 * P * Nk * 64, 77, -1.f, 1.f);
            for (size_t r = 0; r < (size_t)2 * P * Nk; ++r) { double s = 0; for (int c = 0; c < 64; ++c) s += (double)hd[r * 64 + c] * hd[r * 64 + c]; const float inv = (float)(1.0 / std::sqrt(s)); for (int c = 0; c < 64; ++c) hd[r * 64 + c] *= inv; }
            float* d; int64_t *i0, *i1, *w0, *w1; int32_t *nm, *wn; void* ws;
            const size_t wsb = L.match_ws(P, Nk, Nk);
            HIPCHK(hipMalloc(&d, hd.size() * 4)); HIPCHK(hipMemcpy(d, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
            HIPCHK(hipMalloc(&i0, (size_t)P * Nk * 8)); HIPCHK(hipMalloc(&i1, (size_t)P * Nk * 8)); HIPCHK(hipMalloc(&w0, (size_t)P * Nk * 8)); HIPCHK(hipMalloc(&w1, (size_t)P * Nk * 8));
            HIPCHK(hipMalloc(&nm, 64)); HIPCHK(hipMalloc(&wn, 64)); HIPCHK(hipMalloc(&ws, wsb));
            HIPCHK(hipMemset(i0, 0, (size_t)P * Nk * 8)); HIPCHK(hipMemset(i1, 0, (size_t)P * Nk * 8)); HIPCHK(hipMemset(w0, 0, (size_t)P * Nk * 8)); HIPCHK(hipMemset(w1, 0, (size_t)P * Nk * 8));
            HIPCHK(hipMemset(nm, 0, 64)); HIPCHK(hipMemset(wn, 0, 64));
            auto run = [&](int64_t* a0, int64_t* a1, int32_t* an) { return L.match(L.h, d, (size_t)Nk * 64, d + (size_t)P * Nk * 64, (size_t)Nk * 64, nullptr, nullptr, nullptr, nullptr, 0, 0, P, Nk, Nk, -1.f, a0, a1, an, ws, wsb, nullptr); };
            L.cold(0);
            if (run(w0, w1, wn)) printf("match: %s\n", L.last_error());
            HIPCHK(hipDeviceSynchronize()); HIPCHK(hipMemset(total, 0, 4));
            L.cold(1);
            const int nmr = std::max(100, N / 10);
            for (int i = 0; i < nmr; ++i) {
                HIPCHK(hipMemsetAsync(i0, 0, (size_t)P * Nk * 8, nullptr)); HIPCHK(hipMemsetAsync(i1, 0, (size_t)P * Nk * 8, nullptr));      // (rows behind n_matches are unspecified: compare from a common state)
                run(i0, i1, nm);
                cmp_kernel<<<8, 64>>>(reinterpret_cast<const uint4*>(nm), reinterpret_cast<const uint4*>(wn), 4, flag);
                cmp_kernel<<<64, 256>>>(reinterpret_cast<const uint4*>(i0), reinterpret_cast<const uint4*>(w0), (size_t)P * Nk / 2, flag);      // (the lists themselves: rows behind the counts stay zero)
                cmp_kernel<<<64, 256>>>(reinterpret_cast<const uint4*>(i1), reinterpret_cast<const uint4*>(w1), (size_t)P * Nk / 2, flag);
                tally_kernel<<<1, 1>>>(flag, total);
            }
            L.cold(0);
            HIPCHK(hipDeviceSynchronize());
            wrong[nk + 2][pos] = take_total(); runs[nk + 2] = nmr;
            HIPCHK(hipFree(d)); HIPCHK(hipFree(i0)); HIPCHK(hipFree(i1)); HIPCHK(hipFree(w0)); HIPCHK(hipFree(w1)); HIPCHK(hipFree(nm)); HIPCHK(hipFree(wn)); HIPCHK(hipFree(ws));
        }
        {   // the fine_matcher chain: 20 000 rows of 128 features
            const int n = 20000;
            auto hx = rnd((size_t)n * 128, 31, -0.3f, 0.3f);
            float *x, *o, *want; void* ws;
            const size_t wsb = L.refine_ws(1, n);
            HIPCHK(hipMalloc(&x, (size_t)n * 128 * 4)); HIPCHK(hipMalloc(&o, (size_t)n * 64 * 4)); HIPCHK(hipMalloc(&want, (size_t)n * 64 * 4)); HIPCHK(hipMalloc(&ws, wsb));
            HIPCHK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
            L.cold(0);
            if (L.fine(L.h, x, n, want, ws, wsb, nullptr)) printf("fine_matcher: %s\n", L.last_error());
            HIPCHK(hipDeviceSynchronize()); HIPCHK(hipMemset(total, 0, 4));
            L.cold(1);
            const int nf = std::max(200, N / 5);
            for (int i = 0; i < nf; ++i) {
                L.fine(L.h, x, n, o, ws, wsb, nullptr);
                cmp_kernel<<<256, 256>>>(reinterpret_cast<const uint4*>(o), reinterpret_cast<const uint4*>(want), (size_t)n * 16, flag);
                tally_kernel<<<1, 1>>>(flag, total);
            }
            L.cold(0);
            HIPCHK(hipDeviceSynchronize());
            wrong[nk + 3][pos] = take_total(); runs[nk + 3] = nf;
            // xfh_refine_matches on the same numbers (the first layer as linear_fx_kernel<128, gather>): desc0 = columns 0..63 of x's first half of rows, desc1 = the second half,
            // every row matched with itself, fine_conf = -1 (every row kept): the (n / 2, 4) rows and the count against the quiet run
            {
                const int n2 = n / 2;
                std::vector<int64_t> hid(n2); for (int r = 0; r < n2; ++r) hid[r] = (r * 7919) % n2;
                auto hk = rnd((size_t)n2 * 2, 41, 0.f, 600.f);
                std::vector<float> hs(n2, 1.f);
                int64_t *i0, *i1; float *k0, *k1, *sc, *ro, *rw; int32_t *nm, *no, *nw;
                HIPCHK(hipMalloc(&i0, (size_t)n2 * 8)); HIPCHK(hipMalloc(&i1, (size_t)n2 * 8)); HIPCHK(hipMalloc(&k0, (size_t)n2 * 8)); HIPCHK(hipMalloc(&k1, (size_t)n2 * 8)); HIPCHK(hipMalloc(&sc, (size_t)n2 * 4));
                HIPCHK(hipMalloc(&ro, (size_t)n2 * 16)); HIPCHK(hipMalloc(&rw, (size_t)n2 * 16)); HIPCHK(hipMalloc(&nm, 16)); HIPCHK(hipMalloc(&no, 16)); HIPCHK(hipMalloc(&nw, 16));
                HIPCHK(hipMemcpy(i0, hid.data(), (size_t)n2 * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(i1, hid.data(), (size_t)n2 * 8, hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(k0, hk.data(), (size_t)n2 * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(k1, hk.data(), (size_t)n2 * 8, hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy(sc, hs.data(), (size_t)n2 * 4, hipMemcpyHostToDevice));
                const int32_t hn[4] = {n2, 0, 0, 0};
                HIPCHK(hipMemcpy(nm, hn, 16, hipMemcpyHostToDevice)); HIPCHK(hipMemset(no, 0, 16)); HIPCHK(hipMemset(nw, 0, 16));
                HIPCHK(hipMemset(ro, 0, (size_t)n2 * 16)); HIPCHK(hipMemset(rw, 0, (size_t)n2 * 16));
                auto rr = [&](float* out, int32_t* nout) { return L.refine(L.h, x, x + (size_t)n2 * 64, k0, k1, sc, i0, i1, nm, 1, n2, -1.f, out, nout, ws, wsb, nullptr); };
                L.cold(0);
                if (rr(rw, nw)) printf("refine_matches: %s\n", L.last_error());
                HIPCHK(hipDeviceSynchronize()); HIPCHK(hipMemset(total, 0, 4));
                L.cold(1);
                for (int i = 0; i < nf; ++i) {
                    HIPCHK(hipMemsetAsync(ro, 0, (size_t)n2 * 16, nullptr));
                    rr(ro, no);
                    cmp_kernel<<<64, 256>>>(reinterpret_cast<const uint4*>(ro), reinterpret_cast<const uint4*>(rw), (size_t)n2, flag);
                    cmp_kernel<<<1, 64>>>(reinterpret_cast<const uint4*>(no), reinterpret_cast<const uint4*>(nw), 1, flag);
                    tally_kernel<<<1, 1>>>(flag, total);
                }
                L.cold(0);
                HIPCHK(hipDeviceSynchronize());
                wrong[nk + 4][pos] = take_total(); runs[nk + 4] = nf;
                HIPCHK(hipFree(i0)); HIPCHK(hipFree(i1)); HIPCHK(hipFree(k0)); HIPCHK(hipFree(k1)); HIPCHK(hipFree(sc)); HIPCHK(hipFree(ro)); HIPCHK(hipFree(rw)); HIPCHK(hipFree(nm)); HIPCHK(hipFree(no)); HIPCHK(hipFree(nw));
            }
            HIPCHK(hipFree(x)); HIPCHK(hipFree(o)); HIPCHK(hipFree(want)); HIPCHK(hipFree(ws));
        }
        printf("position %2d done\n", pos);
    }
    const char* extra[NEXTRA] = {"block1_mx_kernel<7>", "whole backbone incl. head_bx_kernel<.., fx> x 2 (B = 8)", "xfh_match_mnn: mnn_f16_sweep + refine (match lists of 8 pairs; cold hook at positions 1-15)", "xfh_fine_matcher: linear_fx_kernel<128, fp32> + 3 x linear_fxd_kernel<512> + linear_fx_kernel<512, pair, fp32> (20 000 rows)", "xfh_refine_matches: the same chain behind linear_fx_kernel<128, gather> (10 000 matches)"};
    printf("\ncold-started launches with a result that differs from the quiet one, per code position 0 .. 15 (-1 = library missing):\n");
    for (int ki = (only_fine ? nk + 3 : 0); ki < nk + NEXTRA; ++ki) {
        std::string line; long tot = 0;
        for (int p = 0; p < 16; ++p) { line += " " + std::to_string(wrong[ki][p]); tot += std::max(0l, wrong[ki][p]); }
        printf("%-66s %6ld launches per position:%s   (total wrong %ld)\n", ki < nk ? ks[ki].name : extra[ki - nk], runs[ki], line.c_str(), tot);
    }
    printf("done\n");
    return 0;
}
