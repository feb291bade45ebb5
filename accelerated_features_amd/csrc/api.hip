// C ABI of libxfeat_hip.so (include/xfeat_hip.h): weight packing, workspace planning and the
// launch sequences of the hot path.  No compute happens on the host; there is no CPU fallback.
#include "../../include/xfeat_hip.h"
#include "kernels.hpp"
#include "weight_split.hpp"
#include "block1_fx.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace xfh;

// ------------------------------------------------------------------------------------------
// error reporting
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
// for the other translation units of the library (api_lg.hip)
int xfh_set_error(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return fail(code, "%s", buf);
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(XFH_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(XFH_ERR_HIP, "%s: kernel launch failed: %s", what, hipGetErrorString(e));
    return XFH_OK;
}

// ------------------------------------------------------------------------------------------
// network description (== accelerated_features_amd/spec.py::CONVS, reference model.py:33-111)
// ------------------------------------------------------------------------------------------
struct ConvSpec { int cin, cout, ks, stride, bn; };
static const ConvSpec kConvs[L_NUM] = {
    {1, 24, 1, 1, 0},                                                      // skip1.1
    {1, 4, 3, 1, 1}, {4, 8, 3, 2, 1}, {8, 8, 3, 1, 1}, {8, 24, 3, 2, 1},   // block1
    {24, 24, 3, 1, 1}, {24, 24, 3, 1, 1},                                  // block2
    {24, 64, 3, 2, 1}, {64, 64, 3, 1, 1}, {64, 64, 1, 1, 1},               // block3
    {64, 64, 3, 2, 1}, {64, 64, 3, 1, 1}, {64, 64, 3, 1, 1},               // block4
    {64, 128, 3, 2, 1}, {128, 128, 3, 1, 1}, {128, 128, 3, 1, 1}, {128, 64, 1, 1, 1},   // block5
    {64, 64, 3, 1, 1}, {64, 64, 3, 1, 1}, {64, 64, 1, 1, 0},               // block_fusion
    {64, 64, 1, 1, 1}, {64, 64, 1, 1, 1}, {64, 1, 1, 1, 0},                // heatmap_head
    {64, 64, 1, 1, 1}, {64, 64, 1, 1, 1}, {64, 64, 1, 1, 1}, {64, 65, 1, 1, 0},   // keypoint_head
};
struct FineSpec { int k, n, bn; };
static const FineSpec kFine[5] = {{128, 512, 1}, {512, 512, 1}, {512, 512, 1}, {512, 512, 1}, {512, 64, 0}};
static const double kBnEps = 1e-5;

static int num_arrays() {
    int n = 0;
    for (int i = 0; i < L_NUM; ++i) n += kConvs[i].bn ? 3 : 2;
    for (int i = 0; i < 5; ++i) n += kFine[i].bn ? 4 : 2;
    return n;
}
static size_t array_floats(int idx) {
    int a = 0;
    for (int i = 0; i < L_NUM; ++i) {
        const ConvSpec& c = kConvs[i];
        const int cnt = c.bn ? 3 : 2;
        if (idx < a + cnt) return (idx - a == 0) ? (size_t)c.cout * c.cin * c.ks * c.ks : (size_t)c.cout;
        a += cnt;
    }
    for (int i = 0; i < 5; ++i) {
        const FineSpec& f = kFine[i];
        const int cnt = f.bn ? 4 : 2;
        if (idx < a + cnt) return (idx - a == 0) ? (size_t)f.n * f.k : (size_t)f.n;
        a += cnt;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// profiler: HIP events around one kernel family, on the launch stream
// ------------------------------------------------------------------------------------------
namespace xfh {
int g_debug_cold = 0;      // debug (xfh_debug_cold_start): process-wide, never set by the product path
extern long long* g_head_trace;        // k_heads.hip
struct Profiler {
    int which = 0;
    std::vector<hipEvent_t> ev;   // pairs
    std::vector<int> ids;         // span id of each pair (XFH_PROF_ALL)
    size_t used = 0;
    double flops = 0, bytes = 0;
};
// XFH_PROF_ALL records the leaf spans only (block1, one conv layer, the XFH_SPAN_* ids): the family ids bracket several of them and do not nest
static inline bool prof_on(const Profiler* p, int which) {
    return p && (p->which == which || (p->which == XFH_PROF_ALL && (which == XFH_PROF_BLOCK1 || which >= XFH_PROF_CONV_LAYER0)));
}
void prof_begin(Profiler* p, int which, hipStream_t st) {
    if (!prof_on(p, which)) return;
    if (p->used + 2 > p->ev.size()) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        p->ev.push_back(a);
        p->ev.push_back(b);
    }
    if (p->ids.size() < p->ev.size() / 2) p->ids.resize(p->ev.size() / 2);
    p->ids[p->used / 2] = which;
    (void)hipEventRecord(p->ev[p->used], st);
}
void prof_end(Profiler* p, int which, hipStream_t st, double flops, double bytes) {
    if (!prof_on(p, which)) return;
    (void)hipEventRecord(p->ev[p->used + 1], st);
    p->used += 2;
    p->flops += flops;
    p->bytes += bytes;
}
}  // namespace xfh

struct xfh_context {
    long long* trace = nullptr;   // debug: conv kernel phase stamps
    int device;
    float* blob;          // device weights
    NetWeights nw;
    Profiler prof;
    Options opt;          // xfh_set_option
    int* status = nullptr;      // xfh_set_status_buffer: caller-owned device word the kernels OR status bits into (bit 0: fp16-pair range exceeded)
};

// ------------------------------------------------------------------------------------------
// workspace carving
// ------------------------------------------------------------------------------------------
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base((char*)p) {}
    template <typename T>
    T* take(size_t n) {
        off = align_up(off, 256);
        T* r = base ? (T*)(base + off) : nullptr;
        off += n * sizeof(T);
        return r;
    }
};

struct BackboneWs {
    double* part;
    float* coef;          // per-image instance-norm {alpha, beta}
    float *gray, *x1, *x2a, *x2b, *x3a, *x3b, *x3c, *x4a, *x4b, *x4c, *x5a, *x5b, *x5c, *x5d;
    float *pyr, *f0, *heat_tmp;
};
static size_t carve_backbone(void* ws, int B, int H, int W, BackboneWs& o) {
    Carver c(ws);
    const size_t HW = (size_t)H * W, b = B;
    o.part = c.take<double>(b * GS_CHUNKS * 2);
    o.coef = c.take<float>(b * 2);
    o.gray = c.take<float>(b * HW);
    o.x1 = c.take<float>(b * 24 * HW / 16);
    o.x2a = c.take<float>(b * 24 * HW / 16);
    o.x2b = c.take<float>(b * 24 * HW / 16);
    o.x3a = c.take<float>(b * 64 * HW / 64);
    o.x3b = c.take<float>(b * 64 * HW / 64);
    o.x3c = c.take<float>(b * 64 * HW / 64);
    o.x4a = c.take<float>(b * 64 * HW / 256);
    o.x4b = c.take<float>(b * 64 * HW / 256);
    o.x4c = c.take<float>(b * 64 * HW / 256);
    o.x5a = c.take<float>(b * 128 * HW / 1024);
    o.x5b = c.take<float>(b * 128 * HW / 1024);
    o.x5c = c.take<float>(b * 128 * HW / 1024);
    o.x5d = c.take<float>(b * 64 * HW / 1024);
    o.pyr = c.take<float>(b * 64 * HW / 64);
    o.f0 = c.take<float>(b * 64 * HW / 64);
    o.heat_tmp = c.take<float>(b * HW);
    return align_up(c.off, 256);
}

static size_t carve_detect(void* ws, int B, int H, int W, int top_k, int cap, DetectWs& o) {
    Carver c(ws);
    const size_t WPR = ceil_div(W, 64), b = B;
    o.mask = c.take<unsigned long long>(b * H * WPR);
    o.wcount = c.take<int>(b * H * WPR);
    o.cand = c.take<unsigned>(b * cap);
    o.keys = c.take<unsigned long long>(b * cap);
    o.skeys = c.take<unsigned long long>(b * top_k);
    o.nsel = c.take<int>(b);
    o.invnorm = c.take<float>(b * (H / 8) * (W / 8));
    return align_up(c.off, 256);
}

struct DenseWs { unsigned long long* keys; unsigned long long* skeys; int* nsel; };
static size_t carve_dense(void* ws, int B, int hc, int wc, int k, DenseWs& o) {
    Carver c(ws);
    o.keys = c.take<unsigned long long>((size_t)B * hc * wc);
    o.skeys = c.take<unsigned long long>((size_t)B * k);
    o.nsel = c.take<int>(B);
    return align_up(c.off, 256);
}

static size_t carve_match(void* ws, int P, int N1, int N2, MatchWs& o) {
    Carver c(ws);
    const size_t z0 = c.off;
    o.rowkey = c.take<unsigned long long>((size_t)P * N1);
    o.colkey = c.take<unsigned long long>((size_t)P * N2);
    o.rowmaxh = c.take<unsigned>((size_t)P * N1);
    o.colmaxh = c.take<unsigned>((size_t)P * N2);
    o.nmax = c.take<unsigned>((size_t)2 * P);
    o.zeroed = ws ? (char*)ws + z0 : nullptr;
    o.zeroed_bytes = c.off - z0;
    o.a16 = c.take<_Float16>((size_t)P * N1 * 64);
    o.b16 = c.take<_Float16>((size_t)P * N2 * 64);
    o.na = c.take<float>((size_t)P * N1);
    o.nb = c.take<float>((size_t)P * N2);
    o.thr_row = c.take<float>((size_t)P * N1);
    o.thr_col = c.take<float>((size_t)P * N2);
    o.R = c.take<float>((size_t)P * ceil_div(N2, 32) * N1);      // block maxima: 1/32 of the similarity matrix each
    o.C = c.take<float>((size_t)P * (ceil_div(N1, 1024) * 32) * N2);      // (the one-orientation sweep's blocks: 32 residues per row group of 1024 rows; >= ceil(N1 / 32))
    return align_up(c.off, 256);
}

struct RefineWs { int32_t *offs, *total, *rowmap; float *actA, *actB, *rows; unsigned char* keep; };
static size_t carve_refine(void* ws, int P, int N, RefineWs& o) {
    Carver c(ws);
    const size_t M = (size_t)P * N;
    o.offs = c.take<int32_t>(P + 1);
    o.total = c.take<int32_t>(1);
    o.rowmap = c.take<int32_t>(M);
    o.actA = c.take<float>(M * 512);
    o.actB = c.take<float>(M * 512);
    o.rows = c.take<float>(M * 4);
    o.keep = c.take<unsigned char>(M);
    return align_up(c.off, 256);
}

static int check_ws(const void* ws, size_t have, size_t need) {
    if (!ws) return fail(XFH_ERR_WORKSPACE, "workspace is NULL (need %zu bytes)", need);
    if (((uintptr_t)ws & 255) != 0) return fail(XFH_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    if (have < need) return fail(XFH_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", have, need);
    return XFH_OK;
}


// ------------------------------------------------------------------------------------------
// exported functions
// ------------------------------------------------------------------------------------------
extern "C" {

int xfh_version(void) { return XFH_VERSION; }
const char* xfh_last_error(void) { return g_err; }
int xfh_num_weight_arrays(void) { return num_arrays(); }
size_t xfh_weight_array_floats(int i) { return array_floats(i); }

int xfh_create(const float* const* host_arrays, int n_arrays, int device, xfh_handle* out) {
    if (!host_arrays || !out) return fail(XFH_ERR_ARG, "xfh_create: NULL argument");
    if (n_arrays != num_arrays()) return fail(XFH_ERR_WEIGHTS, "xfh_create: expected %d weight arrays, got %d", num_arrays(), n_arrays);
    for (int i = 0; i < n_arrays; ++i)
        if (!host_arrays[i]) return fail(XFH_ERR_WEIGHTS, "xfh_create: weight array %d is NULL", i);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(XFH_ERR_DEVICE, "xfh_create: no HIP device visible");
    if (device < 0 || device >= ndev) return fail(XFH_ERR_DEVICE, "xfh_create: device %d out of range (%d visible)", device, ndev);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(XFH_ERR_DEVICE, "xfh_create: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    HIP_TRY(hipSetDevice(device));

    std::vector<float> blob;
    auto reserve = [&](size_t n) { size_t o = align_up(blob.size(), 64); blob.resize(o + n, 0.f); return o; };
    const size_t zoff = reserve(256);
    struct Off { size_t oihw, kc, kcp, bias, fx, rs; bool has_fx, has_rs; } coff[L_NUM];
    struct FOff { size_t w, b, fx; bool fx_ok; } foff[5];
    int ai = 0;
    for (int li = 0; li < L_NUM; ++li) {
        const ConvSpec& c = kConvs[li];
        const int kk = c.ks * c.ks, cpad = (c.cout + 31) / 32 * 32;
        const float* w = host_arrays[ai++];
        std::vector<double> scale(c.cout, 1.0), shift(c.cout, 0.0);
        if (c.bn) {
            const float* rm = host_arrays[ai++];
            const float* rv = host_arrays[ai++];
            for (int o = 0; o < c.cout; ++o) {
                if (!(rv[o] + kBnEps > 0)) return fail(XFH_ERR_WEIGHTS, "layer %d: running_var[%d] = %g is not positive", li, o, rv[o]);
                scale[o] = 1.0 / std::sqrt((double)rv[o] + kBnEps);
                shift[o] = -(double)rm[o] * scale[o];
            }
        } else {
            const float* b = host_arrays[ai++];
            for (int o = 0; o < c.cout; ++o) shift[o] = b[o];
        }
        coff[li].oihw = reserve((size_t)c.cout * c.cin * kk);
        coff[li].kc = reserve((size_t)c.cin * kk * c.cout);
        coff[li].kcp = reserve((size_t)c.cin * kk * cpad);
        coff[li].bias = reserve(cpad);
        for (int o = 0; o < c.cout; ++o) {
            for (int i = 0; i < c.cin; ++i)
                for (int t = 0; t < kk; ++t) {
                    const float v = (float)((double)w[((size_t)o * c.cin + i) * kk + t] * scale[o]);
                    blob[coff[li].oihw + ((size_t)o * c.cin + i) * kk + t] = v;
                    blob[coff[li].kc + ((size_t)i * kk + t) * c.cout + o] = v;
                    blob[coff[li].kcp + ((size_t)i * kk + t) * cpad + o] = v;
                }
            blob[coff[li].bias + o] = (float)shift[o];
        }
        // fp16-pair MFMA paths (k_conv_bx.hip, k_conv_bx64s2x.hip, k_conv_rs64.hip): every fp32 weight as three fp16 fragments (weight_split.hpp: split_weight) in MFMA
        // operand order -- only if every |w| of the layer stays below kFxMaxWeight (the layer otherwise runs on the f32-MFMA kernel).
        // K group kg = 2 step + half = (tap, 8-channel group) for the 24-channel layers.
        const bool bx24 = c.ks == 3 && c.stride == 1 && c.cin == 24 && c.cout <= 32;
        const bool bx64 = c.ks == 3 && c.stride == 1 && c.cin == 64 && c.cout == 64;
        const bool bx24s2 = c.ks == 3 && c.stride == 2 && c.cin == 24 && c.cout == 64;
        const bool bx1x1 = c.ks == 1 && c.cin == 64 && c.cout == 64 && li > 0 && kConvs[li - 1].ks == 3 && kConvs[li - 1].cout == 64 && kConvs[li - 1].cin == 64;      // block3.2, block_fusion.2
        const bool bx64s2 = c.ks == 3 && c.stride == 2 && c.cin == 64 && (c.cout == 64 || c.cout == 128);      // block4.0, block5.0
        const bool bx128 = c.ks == 3 && c.stride == 1 && c.cin == 128 && c.cout == 128;      // block5.1, block5.2
        bool fx_ok = bx24 || bx64 || bx24s2 || bx1x1 || bx64s2 || bx128;
        if (fx_ok) {
            float wmax = 0.f;
            for (size_t i = 0; i < (size_t)c.cout * c.cin * kk; ++i) wmax = std::max(wmax, std::fabs(blob[coff[li].oihw + i]));
            fx_ok = wmax < kFxMaxWeight;
        }
        coff[li].has_fx = fx_ok && (bx24 || bx24s2 || bx64s2);
        if (coff[li].has_fx) {
            size_t words = 0;
            const int cg = c.cin / 8, nstep = bx_steps(c.cin), nch = c.cin / 16, nhf = c.cout / 64;
            if (bx24) words = (size_t)nstep * 3 * 64 * 4;
            if (bx24s2) words = (size_t)2 * nstep * 3 * 64 * 4;
            if (bx64s2) words = (size_t)nhf * nch * 9 * 2 * 3 * 64 * 4;
            coff[li].fx = reserve(words);
            uint16_t* dst = reinterpret_cast<uint16_t*>(&blob[coff[li].fx]);
            uint16_t q[3];
            if (bx24)         // [step][fragment][lane = half * 32 + cout][8]
                for (int st = 0; st < nstep; ++st)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int o = lane & 31, kg = 2 * st + (lane >> 5);
                        for (int i = 0; i < 8; ++i) {
                            float v = 0.f;
                            if (o < c.cout && kg < 9 * cg) v = blob[coff[li].oihw + ((size_t)o * c.cin + (kg % cg) * 8 + i) * 9 + kg / cg];
                            split_weight(v, q);
                            for (int sp = 0; sp < 3; ++sp) dst[(((size_t)st * 3 + sp) * 64 + lane) * 8 + i] = q[sp];
                        }
                    }
            if (bx24s2)       // the stride-2 sibling: [cout block][step][fragment][lane][8], same K order as bx24
                for (int cb = 0; cb < 2; ++cb)
                    for (int st = 0; st < nstep; ++st)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int o = cb * 32 + (lane & 31), kg = 2 * st + (lane >> 5);
                            for (int i = 0; i < 8; ++i) {
                                float v = 0.f;
                                if (kg < 9 * cg) v = blob[coff[li].oihw + ((size_t)o * c.cin + (kg % cg) * 8 + i) * 9 + kg / cg];
                                split_weight(v, q);
                                for (int sp = 0; sp < 3; ++sp) dst[((((size_t)cb * nstep + st) * 3 + sp) * 64 + lane) * 8 + i] = q[sp];
                            }
                        }
            if (bx64s2) pack_bx64(&blob[coff[li].oihw], c.cin, c.cout, dst);      // (weight_split.hpp)
        }
        coff[li].has_rs = false;
        if (bx1x1 && fx_ok) {      // the 1x1 behind a 64 -> 64 3x3 in conv_rs64_kernel's order (16 couts per wave, natural K order)
            coff[li].rs = reserve((size_t)4 * 2 * 3 * 64 * 4);
            pack_rs64_1x1(&blob[coff[li].oihw], reinterpret_cast<uint16_t*>(&blob[coff[li].rs]));
            coff[li].has_rs = true;
        }
        if (bx128 && fx_ok) {      // block5.1, block5.2: conv_rs64_kernel's 128-channel form
            coff[li].rs = reserve(4 * kRs64Halfs / 2);
            pack_rs128(&blob[coff[li].oihw], reinterpret_cast<uint16_t*>(&blob[coff[li].rs]));
            coff[li].has_rs = true;
        }
        if (bx64 && fx_ok) {      // conv_rs64_kernel's order (one K quarter per wave)
            coff[li].rs = reserve(kRs64Halfs / 2);
            pack_rs64(&blob[coff[li].oihw], reinterpret_cast<uint16_t*>(&blob[coff[li].rs]));
            coff[li].has_rs = true;
        }
    }
    // heads on the fp16 matrix cores (k_heads.hip: head_bx_kernel): per layer [K step t][cout block][fragment][lane = half * 32 + cout][8].
    // K order: the first layer takes its channels in natural order (16 t + 8 half + i); a chained layer takes the previous layer's D
    // registers, i.e. feature 32 (t >> 1) + 16 (t & 1) + 8 (i >> 2) + 4 half + (i & 3).
    // (the images hold 64 of keypoint_head.3's 65 outputs: the dustbin logit is a dot product on the vector ALUs -- head_bx_body.hpp)
    size_t head_off[2] = {0, 0}, head_boff[2] = {0, 0};      // [head]
    bool head_fx_ok[2] = {true, true};
    const float head_b_last = blob[coff[L_HEAT_2].bias];
    {
        const int kp[4] = {L_KP_0, L_KP_1, L_KP_2, L_KP_3}, rel[2] = {L_HEAT_0, L_HEAT_1};
        for (int hd = 0; hd < 2; ++hd) {
            const int nl = hd == 0 ? 4 : 2;
            const int* ls = hd == 0 ? kp : rel;
            size_t words = 0, nbias = 0;
            auto couts = [&](int p) { return hd == 0 && p == 3 ? 64 : kConvs[ls[p]].cout; };
            for (int p = 0; p < nl; ++p) { words += (size_t)4 * ((couts(p) + 31) / 32) * 3 * 64 * 4; nbias += 32 * ((kConvs[ls[p]].cout + 31) / 32); }
            head_off[hd] = reserve(words);
            head_boff[hd] = reserve(nbias);
            uint16_t* dst = reinterpret_cast<uint16_t*>(&blob[head_off[hd]]);
            size_t bo = head_boff[hd];
            for (int p = 0; p < nl; ++p) {
                const ConvSpec& c = kConvs[ls[p]];
                const int mbo = (c.cout + 31) / 32;
                for (size_t i = 0; i < (size_t)c.cout * 64; ++i)
                    if (!(std::fabs(blob[coff[ls[p]].oihw + i]) < kFxMaxWeight)) head_fx_ok[hd] = false;
                dst += pack_head_layer(&blob[coff[ls[p]].oihw], couts(p), p == 0, dst);      // (weight_split.hpp)
                for (int o = 0; o < 32 * mbo; ++o) blob[bo + o] = o < c.cout ? blob[coff[ls[p]].bias + o] : 0.f;
                bo += 32 * mbo;
            }
        }
    }
    // block1.3 (8 -> 24, stride 2) for block1_mx_kernel: the compact fp16-pair image of block1_fx.hpp (only if every |w| stays below kFxMaxWeight)
    size_t b1fx_off = 0;
    bool b1fx_ok = true;
    {
        const float* wkc = &blob[coff[L_BLOCK1_3].kc];
        for (int i = 0; i < 8 * 9 * 24; ++i) b1fx_ok = b1fx_ok && std::fabs(wkc[i]) < kFxMaxWeight;
        b1fx_off = reserve(b1fx::W4_BYTES / 4);
        b1fx::pack_w4(&blob[coff[L_BLOCK1_3].kc], reinterpret_cast<uint16_t*>(&blob[b1fx_off]), [](float v, uint16_t (&q)[3]) { split_weight(v, q); });
    }
    size_t b1fx3_off = 0;      // block1.2 (8 -> 8) likewise
    {
        const float* wkc = &blob[coff[L_BLOCK1_2].kc];
        for (int i = 0; i < 8 * 9 * 8; ++i) b1fx_ok = b1fx_ok && std::fabs(wkc[i]) < kFxMaxWeight;
        b1fx3_off = reserve(b1fx::W3_IMAGE_BYTES / 4);
        b1fx::pack_w3(&blob[coff[L_BLOCK1_2].kc], reinterpret_cast<uint16_t*>(&blob[b1fx3_off]), [](float v, uint16_t (&q)[3]) { split_weight(v, q); });
    }
    for (int fi = 0; fi < 5; ++fi) {
        const FineSpec& f = kFine[fi];
        const int npad = (f.n + 63) / 64 * 64;
        const float* w = host_arrays[ai++];
        const float* b = host_arrays[ai++];
        std::vector<double> scale(f.n, 1.0), shift(f.n, 0.0);
        if (f.bn) {
            const float* rm = host_arrays[ai++];
            const float* rv = host_arrays[ai++];
            for (int o = 0; o < f.n; ++o) {
                if (!(rv[o] + kBnEps > 0)) return fail(XFH_ERR_WEIGHTS, "fine_matcher %d: running_var[%d] not positive", fi, o);
                scale[o] = 1.0 / std::sqrt((double)rv[o] + kBnEps);
                shift[o] = ((double)b[o] - (double)rm[o]) * scale[o];
            }
        } else {
            for (int o = 0; o < f.n; ++o) shift[o] = b[o];
        }
        foff[fi].w = reserve((size_t)f.k * npad);
        foff[fi].b = reserve(npad);
        for (int o = 0; o < f.n; ++o) {
            for (int i = 0; i < f.k; ++i) blob[foff[fi].w + (size_t)i * npad + o] = (float)((double)w[(size_t)o * f.k + i] * scale[o]);
            blob[foff[fi].b + o] = (float)shift[o];
        }
        // the same layer as fp16-pair fragments for linear_fx_kernel (only if every |w| stays below kFxMaxWeight: 2^11 w must be an fp16 number)
        float wmax = 0.f;
        for (size_t i = 0; i < (size_t)f.k * npad; ++i) wmax = std::fmax(wmax, std::fabs(blob[foff[fi].w + i]));
        foff[fi].fx_ok = wmax < kFxMaxWeight && f.k % 32 == 0;
        foff[fi].fx = 0;
        if (foff[fi].fx_ok) {
            foff[fi].fx = reserve(((size_t)3 * f.k * npad + 1) / 2);
            std::vector<float> wk(blob.begin() + foff[fi].w, blob.begin() + foff[fi].w + (size_t)f.k * npad);      // (reserve may have moved the blob)
            pack_linear_fx(wk.data(), f.k, npad, reinterpret_cast<uint16_t*>(blob.data() + foff[fi].fx));
        }
    }

    xfh_context* ctx = new xfh_context();
    ctx->device = device;
    ctx->blob = nullptr;
    hipError_t e = hipMalloc((void**)&ctx->blob, blob.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(ctx->blob, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (ctx->blob) (void)hipFree(ctx->blob);
        delete ctx;
        return fail(XFH_ERR_HIP, "xfh_create: weight upload failed: %s", hipGetErrorString(e));
    }
    for (int li = 0; li < L_NUM; ++li) {
        const ConvSpec& c = kConvs[li];
        ConvW& w = ctx->nw.conv[li];
        w.cin = c.cin; w.cout = c.cout; w.ks = c.ks; w.stride = c.stride; w.relu = c.bn; w.cout_pad = (c.cout + 31) / 32 * 32;
        w.w_oihw = ctx->blob + coff[li].oihw;
        w.w_kc = ctx->blob + coff[li].kc;
        w.w_kcp = ctx->blob + coff[li].kcp;
        w.bias = ctx->blob + coff[li].bias;
        w.w_fx = coff[li].has_fx ? ctx->blob + coff[li].fx : nullptr;
        w.w_rs = coff[li].has_rs ? ctx->blob + coff[li].rs : nullptr;
    }
    ctx->nw.zeros = ctx->blob + zoff;
    for (int hd = 0; hd < 2; ++hd) {
        ctx->nw.head_fx[hd] = head_fx_ok[hd] ? ctx->blob + head_off[hd] : nullptr;
        ctx->nw.head_bx_bias[hd] = ctx->blob + head_boff[hd];
    }
    ctx->nw.block1_fx = b1fx_ok ? ctx->blob + b1fx_off : nullptr;
    ctx->nw.block1_fx3 = b1fx_ok ? ctx->blob + b1fx3_off : nullptr;
    ctx->nw.head_rel_b_last = head_b_last;
    ctx->nw.head_kp_b_dust = blob[coff[L_KP_3].bias + 64];
    for (int fi = 0; fi < 5; ++fi) {
        LinW& l = ctx->nw.fine[fi];
        l.k = kFine[fi].k; l.n = kFine[fi].n; l.n_pad = (kFine[fi].n + 63) / 64 * 64; l.relu = kFine[fi].bn;
        l.w_kn = ctx->blob + foff[fi].w;
        l.bias = ctx->blob + foff[fi].b;
        l.w_fx = foff[fi].fx_ok ? ctx->blob + foff[fi].fx : nullptr;
    }
    *out = ctx;
    return XFH_OK;
}

void xfh_destroy(xfh_handle h) {
    if (!h) return;
    for (hipEvent_t e : h->prof.ev) (void)hipEventDestroy(e);
    if (h->blob) (void)hipFree(h->blob);
    delete h;
}

int xfh_resize_bilinear(const float* src, int planes, int Hin, int Win, float* dst, int Hout, int Wout, float scale_h,
                        float scale_w, xfh_stream stream) {
    if (!src || !dst || planes <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0)
        return fail(XFH_ERR_ARG, "xfh_resize_bilinear: bad argument");
    if (planes > 65535) return fail(XFH_ERR_ARG, "xfh_resize_bilinear: more than 65535 planes");
    launch_resize_bilinear(src, planes, Hin, Win, dst, Hout, Wout, scale_h, scale_w, (hipStream_t)stream);
    return check_launch("xfh_resize_bilinear");
}

static int check_img(const char* fn, int B, int C, int H, int W) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return fail(XFH_ERR_ARG, "%s: non-positive dimension", fn);
    if (H % 32 || W % 32) return fail(XFH_ERR_ARG, "%s: H and W must be multiples of 32 (got %dx%d)", fn, H, W);
    if (H >= 65536 || W >= 65536) return fail(XFH_ERR_ARG, "%s: image side >= 65536", fn);
    if (B > 65535) return fail(XFH_ERR_ARG, "%s: batch > 65535", fn);
    return XFH_OK;
}

size_t xfh_backbone_workspace_bytes(int B, int C, int H, int W) {
    (void)C;
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    BackboneWs o;
    return carve_backbone(nullptr, B, H, W, o);
}

static int conv_mfma_checked(xfh_handle h, int layer, int fused_layer, const float* in, int B, int Hin, int Win, float* out,
                             bool nhwc, hipStream_t st, bool in_backbone = false, int link = 0) {      // link (backbone only): bit 0 = the input, bit 1 = the output is a channels-last link between two fp16-pair 24-channel layers      // in_backbone: the layer's neighbours are this call's (the split-format link may be used)
    const ConvW& c = h->nw.conv[layer];
    const ConvW* c2 = fused_layer >= 0 ? &h->nw.conv[fused_layer] : nullptr;
    const int pad = c.ks / 2;
    const int Hout = (Hin + 2 * pad - c.ks) / c.stride + 1, Wout = (Win + 2 * pad - c.ks) / c.stride + 1;
    // which >= 100 selects one layer (100 + layer index), otherwise the whole family
    int pid = h->prof.which >= 100 ? 100 + layer : XFH_PROF_CONV_MFMA;      // (XFH_PROF_ALL = 1000: one span per layer)
    if (h->prof.which == XFH_PROF_CONV_64_64_S1 && c.cin == 64 && c.cout == 64 && c.ks == 3 && c.stride == 1) pid = XFH_PROF_CONV_64_64_S1;
    if (h->prof.which == XFH_PROF_CONV_24_24 && c.cin == 24 && c.cout == 24 && c.ks == 3 && c.stride == 1) pid = XFH_PROF_CONV_24_24;
    prof_begin(&h->prof, pid, st);
    // The fp16-pair kernels first (option fx: bit 1 = the 64- and 128-channel layers, bit 2 = the 24-channel layers); a layer they do not take -- the bit is off, a weight
    // beyond the pair's range left the layer without its image, a map beyond a kernel's exact range -- runs on the f32-MFMA kernel (fp32's range: what the range guard's
    // re-run uses).  Every choice is by layer and image size alone, never by the batch: an image's features do not depend on the batch it travels in.
    int rc = -1;
    const int fx = h->opt.fx;
    if ((fx & XFH_FX_CONV64) && c.ks == 3 && c.stride == 1 && c.w_rs) {
        if (c.cin == 64 && !c2 && !nhwc) rc = launch_conv_rs64(c, in, B, Hin, Win, out, st, h->status, nullptr, false, h->trace);          // block4.1, block4.2, block_fusion.0
        else if (c.cin == 128 && !c2 && !nhwc) rc = launch_conv_rs128(c, in, B, Hin, Win, out, st, h->status);                           // block5.1, block5.2
        else if (c.cin == 64 && c2 && c2->w_rs) rc = launch_conv_rs64(c, in, B, Hin, Win, out, st, h->status, c2, nhwc, h->trace);        // block3.1 + .2, block_fusion.1 + .2
    }
    if (rc && (fx & XFH_FX_CONV64) && c.w_fx && !c2 && !nhwc && c.stride == 2 && c.cin == 64) rc = launch_conv_bx64s2_fx(c, in, B, Hin, Win, out, st, h->trace, h->status);      // block4.0, block5.0
    if (rc && (fx & XFH_FX_CONV24) && c.w_fx && !c2 && !nhwc && c.cin == 24) rc = launch_conv_bx(c, in, B, Hin, Win, out, st, h->trace, h->status, link & 1, (link & 2) != 0);      // block2.0, block2.1, block3.0
    if (rc && link) return fail(XFH_ERR_UNSUPPORTED, "layer %d: a channels-last link without its fp16-pair kernel", layer);      // (cannot happen: the links are chosen by the same conditions)
    if (rc) rc = launch_conv_mfma(c, c2, h->nw.zeros, in, B, Hin, Win, out, nhwc, st, h->trace);
    const int cl = c2 ? c2->cout : c.cout;
    double bytes = 4.0 * ((double)B * c.cin * Hin * Win + (double)B * cl * Hout * Wout + (double)c.cin * c.cout * c.ks * c.ks);
    double flops = conv_flops(c, B, Hout, Wout);
    if (c2) { flops += conv_flops(*c2, B, Hout, Wout); bytes += 4.0 * c2->cin * c2->cout; }
    prof_end(&h->prof, pid, st, flops, bytes);
    if (rc) return fail(XFH_ERR_UNSUPPORTED, "no MFMA conv instantiation for layer %d (%d->%d k%d s%d) at %dx%d", layer, c.cin, c.cout, c.ks, c.stride, Hin, Win);
    return XFH_OK;
}

struct ResizeSpec { int Hin, Win, Hm, Wm; float s1h, s1w, s2h, s2w; };      // xfh_backbone_resized: img is (B,C,Hin,Win)

static int backbone_impl(xfh_handle h, const float* img, const unsigned char* img_u8, int u8_layout, float u8_divisor, int B, int C, int H,
                         int W, float* feats, float* logits, float* heat, float* reliab, float* invnorm, void* workspace, size_t workspace_bytes,
                         xfh_stream stream, const ResizeSpec* rs = nullptr) {
    if (!h || (!img && !img_u8) || !feats || !reliab) return fail(XFH_ERR_ARG, "xfh_backbone: NULL argument");
    if (!logits && !heat) return fail(XFH_ERR_ARG, "xfh_backbone: logits and heat are both NULL");
    int rc = check_img("xfh_backbone", B, C, H, W);
    if (rc) return rc;
    BackboneWs w;
    const size_t need = carve_backbone(workspace, B, H, W, w);
    if ((rc = check_ws(workspace, workspace_bytes, need))) return rc;
    hipStream_t st = (hipStream_t)stream;
    const NetWeights& nw = h->nw;
    const int H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8, H16 = H / 16, W16 = W / 16, H32 = H / 32, W32 = W / 32;

    prof_begin(&h->prof, XFH_SPAN_GRAY, st);
    if (rs) {
        if (launch_gray_norm_resized(img, B, C, rs->Hin, rs->Win, rs->Hm, rs->Wm, rs->s1h, rs->s1w, H, W, rs->s2h, rs->s2w, w.part, w.gray,
                                     w.coef, st, h->opt.resize2))
            return fail(XFH_ERR_UNSUPPORTED, "xfh_backbone_resized: second resize step (%g, %g) must be below 2", rs->s2h, rs->s2w);
    } else if (img_u8) launch_gray_norm_u8(img_u8, u8_layout == XFH_LAYOUT_NHWC, u8_divisor, B, C, H, W, w.part, w.gray, w.coef, st);
    else launch_gray_norm(img, B, C, H, W, w.part, w.gray, w.coef, st);
    prof_end(&h->prof, XFH_SPAN_GRAY, st, 0, 0);
    prof_begin(&h->prof, XFH_PROF_BLOCK1, st);
    launch_block1_fused(nw, w.gray, w.coef, B, H, W, w.x1, st, h->opt.block1, h->status);
    // block1 + skip1 per input pixel: conv1 9*4*2 + conv2 36*8*2/4 + conv3 72*8*2/4 + conv4 72*24*2/16 = 720 FLOP; gray in, x1 out: 10 bytes
    prof_end(&h->prof, XFH_PROF_BLOCK1, st, 720.0 * B * H * W, 10.0 * B * H * W);
#define CONV(layer, fused, in, hin, win, out, nhwc) \
    if ((rc = conv_mfma_checked(h, layer, fused, in, B, hin, win, out, nhwc, st, true))) return rc
    {   // block2.0 -> block2.1 -> block3.0: channels-last between them when all three run on their fp16-pair kernels (16-byte loads and stores: DESIGN 3.9); planes otherwise
        const bool cl = (h->opt.fx & XFH_FX_CONV24) && conv_bx_links(nw.conv[L_BLOCK2_0], h->trace != nullptr) && conv_bx_links(nw.conv[L_BLOCK2_1], h->trace != nullptr) &&
                        conv_bx_links(nw.conv[L_BLOCK3_0], h->trace != nullptr) && (size_t)24 * H4 * W4 * sizeof(float) < 0x7fffffffu;
#define CONVL(layer, in, out, link) if ((rc = conv_mfma_checked(h, layer, -1, in, B, H4, W4, out, false, st, true, link))) return rc
        CONVL(L_BLOCK2_0, w.x1, w.x2a, cl ? 2 : 0);
        CONVL(L_BLOCK2_1, w.x2a, w.x2b, cl ? 3 : 0);
        CONVL(L_BLOCK3_0, w.x2b, w.x3a, cl ? 1 : 0);
#undef CONVL
    }
    CONV(L_BLOCK3_1, L_BLOCK3_2, w.x3a, H8, W8, w.x3c, false);        // 3x3 + fused 1x1
    CONV(L_BLOCK4_0, -1, w.x3c, H8, W8, w.x4a, false);
    CONV(L_BLOCK4_1, -1, w.x4a, H16, W16, w.x4b, false);
    CONV(L_BLOCK4_2, -1, w.x4b, H16, W16, w.x4c, false);
    CONV(L_BLOCK5_0, -1, w.x4c, H16, W16, w.x5a, false);
    CONV(L_BLOCK5_1, -1, w.x5a, H32, W32, w.x5b, false);
    bool pyr_done = false;
    if ((h->opt.fx & XFH_FX_CONV64) && nw.conv[L_BLOCK5_2].w_rs && conv_rs128_fits(W32)) {
        CONV(L_BLOCK5_2, -1, w.x5b, H32, W32, w.x5a, false);          // conv_rs64_kernel's 128-channel form holds a quarter of the couts per workgroup: the 1x1 (128 -> 64) is not fused into it (x5a is free since block5.1) ...
        prof_begin(&h->prof, XFH_SPAN_PYRAMID, st);
        pyr_done = launch_pyramid53(nw.conv[L_BLOCK5_3], w.x3c, w.x4c, w.x5a, w.pyr, B, H8, W8, H16, W16, H32, W32, st) == 0;      // ... but into the pyramid sum: x5 never reaches HBM
        if (pyr_done) prof_end(&h->prof, XFH_SPAN_PYRAMID, st, 0, 0);
        else CONV(L_BLOCK5_3, -1, w.x5a, H32, W32, w.x5d, false);      // (the span begun above is simply begun again below: prof_begin alone records nothing)
    } else
    CONV(L_BLOCK5_2, L_BLOCK5_3, w.x5b, H32, W32, w.x5d, false);      // 3x3 + fused 1x1 (128->64)
    if (!pyr_done) {
        prof_begin(&h->prof, XFH_SPAN_PYRAMID, st);
        launch_pyramid_sum(w.x3c, w.x4c, w.x5d, w.pyr, B * 64, H8, W8, H16, W16, H32, W32, st);
        prof_end(&h->prof, XFH_SPAN_PYRAMID, st, 0, 0);
    }
    CONV(L_FUSION_0, -1, w.pyr, H8, W8, w.f0, false);
    CONV(L_FUSION_1, L_FUSION_2, w.f0, H8, W8, feats, true);          // 3x3 + fused 1x1 -> channels-last M1
#undef CONV

    // fused heads: reliability from the channels-last features, key-point head from the gray image
    const bool all = h->prof.which == XFH_PROF_ALL;      // one span per head instead of one for both
    prof_begin(&h->prof, all ? XFH_SPAN_HEAD_REL : XFH_PROF_HEADS, st);
    const bool heads_f32 = !(h->opt.fx & XFH_FX_HEADS);
    launch_rel_head(nw, feats, B * H8 * W8, reliab, invnorm, st, heads_f32, h->status);
    if (all) { prof_end(&h->prof, XFH_SPAN_HEAD_REL, st, 0, 0); prof_begin(&h->prof, XFH_SPAN_HEAD_KP, st); }
    launch_kp_head(nw, w.gray, w.coef, B, H, W, heat ? heat : w.heat_tmp, logits, st, heads_f32, h->status);
    prof_end(&h->prof, all ? XFH_SPAN_HEAD_KP : XFH_PROF_HEADS, st, 0, 0);
    return check_launch("xfh_backbone");
}

int xfh_backbone(xfh_handle h, const float* img, int B, int C, int H, int W, float* feats, float* logits, float* heat,
                 float* reliab, float* invnorm, void* workspace, size_t workspace_bytes, xfh_stream stream) {
    return backbone_impl(h, img, nullptr, 0, 1.f, B, C, H, W, feats, logits, heat, reliab, invnorm, workspace, workspace_bytes, stream);
}

int xfh_backbone_u8(xfh_handle h, const uint8_t* img, int layout, float divisor, int B, int C, int H, int W, float* feats,
                    float* logits, float* heat, float* reliab, float* invnorm, void* workspace, size_t workspace_bytes, xfh_stream stream) {
    if (layout != XFH_LAYOUT_NCHW && layout != XFH_LAYOUT_NHWC) return fail(XFH_ERR_ARG, "xfh_backbone_u8: layout must be XFH_LAYOUT_NCHW or XFH_LAYOUT_NHWC");
    if (!(divisor > 0.f)) return fail(XFH_ERR_ARG, "xfh_backbone_u8: divisor must be positive");
    return backbone_impl(h, nullptr, img, layout, divisor, B, C, H, W, feats, logits, heat, reliab, invnorm, workspace, workspace_bytes, stream);
}

int xfh_backbone_resized(xfh_handle h, const float* img, int B, int C, int Hin, int Win, int Hmid, int Wmid, float scale1_h, float scale1_w,
                         int Hout, int Wout, float scale2_h, float scale2_w, float* feats, float* logits, float* heat, float* reliab, float* invnorm,
                         void* workspace, size_t workspace_bytes, xfh_stream stream) {
    if (Hin <= 0 || Win <= 0 || Hmid <= 0 || Wmid <= 0 || !(scale1_h > 0.f) || !(scale1_w > 0.f))
        return fail(XFH_ERR_ARG, "xfh_backbone_resized: bad source / intermediate size");
    const ResizeSpec rs{Hin, Win, Hmid, Wmid, scale1_h, scale1_w, scale2_h, scale2_w};
    return backbone_impl(h, img, nullptr, 0, 1.f, B, C, Hout, Wout, feats, logits, heat, reliab, invnorm, workspace, workspace_bytes, stream, &rs);
}

int xfh_conv_layer(xfh_handle h, int layer, const float* in, int B, int Hin, int Win, float* out, int variant,
                   xfh_stream stream) {
    if (!h || !in || !out) return fail(XFH_ERR_ARG, "xfh_conv_layer: NULL argument");
    if (layer < 0 || layer >= L_NUM) return fail(XFH_ERR_ARG, "xfh_conv_layer: layer %d out of range", layer);
    if (B <= 0 || Hin <= 0 || Win <= 0 || B > 4000) return fail(XFH_ERR_ARG, "xfh_conv_layer: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const ConvW& c = h->nw.conv[layer];
    if (variant == XFH_CONV_VARIANT_GENERIC) {
        launch_conv_generic(c, in, B, Hin, Win, out, st);
        return check_launch("xfh_conv_layer(generic)");
    }
    if (variant == XFH_CONV_VARIANT_FX) {      // the layer's fp16-pair kernel, or XFH_ERR_UNSUPPORTED -- no silent fallback (tests pin this kernel against the others)
        int rc = -1;
        if (c.ks == 3 && c.cin == 24) rc = launch_conv_bx(c, in, B, Hin, Win, out, st, h->trace, h->status);
        else if (c.ks == 3 && c.cin == 64 && c.stride == 2) rc = launch_conv_bx64s2_fx(c, in, B, Hin, Win, out, st, h->trace, h->status);
        else if (c.ks == 3 && c.cin == 64 && c.stride == 1 && c.w_rs) rc = launch_conv_rs64(c, in, B, Hin, Win, out, st, h->status, nullptr, false, h->trace);
        else if (c.ks == 3 && c.cin == 128 && c.stride == 1 && c.w_rs) rc = launch_conv_rs128(c, in, B, Hin, Win, out, st, h->status);
        if (rc) return fail(XFH_ERR_UNSUPPORTED, "xfh_conv_layer: no fp16-pair kernel for layer %d at %dx%d", layer, Hin, Win);
        return check_launch("xfh_conv_layer(fp16 pair)");
    }
    if (variant == XFH_CONV_VARIANT_FX_PAIR || variant == XFH_CONV_VARIANT_FX_PAIR_NHWC) {      // a 3x3 + the 1x1 behind it as ONE launch of conv_rs64_kernel (block3.1 + .2, block_fusion.1 + .2): NCHW / channels-last output
        if (layer + 1 >= L_NUM) return fail(XFH_ERR_ARG, "xfh_conv_layer: layer %d has no successor", layer);
        const ConvW& c2 = h->nw.conv[layer + 1];
        if (c2.ks != 1 || c2.cin != 64 || c2.cout != 64 || c.cin != 64 || c.ks != 3 || c.stride != 1 || !c.w_rs || !c2.w_rs ||
            launch_conv_rs64(c, in, B, Hin, Win, out, st, h->status, &c2, variant == XFH_CONV_VARIANT_FX_PAIR_NHWC, nullptr))
            return fail(XFH_ERR_UNSUPPORTED, "xfh_conv_layer: no fused 3x3 + 1x1 kernel for layers %d, %d at width %d", layer, layer + 1, Win);
        return check_launch("xfh_conv_layer(3x3 + 1x1)");
    }
    if (variant == XFH_CONV_VARIANT_F32) {      // the f32-MFMA kernel (the range fallback) of the layer
        if (launch_conv_mfma(c, nullptr, h->nw.zeros, in, B, Hin, Win, out, false, st, h->trace))
            return fail(XFH_ERR_UNSUPPORTED, "xfh_conv_layer: no f32-MFMA kernel for layer %d at %dx%d", layer, Hin, Win);
        return check_launch("xfh_conv_layer(f32 mfma)");
    }
    if (variant != 0) return fail(XFH_ERR_ARG, "xfh_conv_layer: unknown variant %d", variant);
    if (layer >= L_BLOCK1_0 && layer <= L_BLOCK1_3) {
        launch_block1_layer(h->nw, layer, in, B, Hin, Win, out, st);
        return check_launch("xfh_conv_layer(block1)");
    }
    if (layer == L_SKIP1 || layer >= L_HEAT_0)
        return fail(XFH_ERR_UNSUPPORTED, "xfh_conv_layer: layer %d runs fused / channels-last in the backbone; use variant 1", layer);
    int rc = conv_mfma_checked(h, layer, -1, in, B, Hin, Win, out, false, st);
    if (rc) return rc;
    return check_launch("xfh_conv_layer(mfma)");
}

size_t xfh_detect_workspace_bytes(int B, int H, int W, int top_k, int nms_capacity) {
    if (B <= 0 || H <= 0 || W <= 0 || top_k <= 0 || nms_capacity <= 0) return 0;
    DetectWs o;
    return carve_detect(nullptr, B, H, W, top_k, nms_capacity, o);
}

int xfh_detect_sparse(xfh_handle h, const float* heat, const float* reliab, const float* feats, const float* invnorm, int B, int H, int W,
                      float threshold, int top_k, int nms_capacity, float rw, float rh, float* kpts, float* scores,
                      float* desc, uint16_t* desc_f16, int32_t* n_valid, int32_t* n_candidates, void* workspace, size_t workspace_bytes,
                      xfh_stream stream) {
    if (!h || !heat || !reliab || !feats || !kpts || !scores || !desc || !n_valid || !n_candidates)
        return fail(XFH_ERR_ARG, "xfh_detect_sparse: NULL argument");
    int rc = check_img("xfh_detect_sparse", B, 1, H, W);
    if (rc) return rc;
    if (top_k <= 0) return fail(XFH_ERR_ARG, "xfh_detect_sparse: top_k %d must be positive", top_k);
    if ((long)(H / 8) * (W / 8) >= (1L << 22))      // descriptor_kernel addresses the feature map with 32-bit byte offsets
        return fail(XFH_ERR_UNSUPPORTED, "xfh_detect_sparse: %dx%d is beyond 2^22 feature cells (about 16k x 16k pixels)", H, W);
    if (nms_capacity <= 0 || (long)nms_capacity > (long)H * W) return fail(XFH_ERR_ARG, "xfh_detect_sparse: nms_capacity %d outside 1..H*W", nms_capacity);
    DetectWs w;
    const size_t need = carve_detect(workspace, B, H, W, top_k, nms_capacity, w);
    if ((rc = check_ws(workspace, workspace_bytes, need))) return rc;
    launch_detect(w, heat, reliab, feats, invnorm, B, H, W, threshold, top_k, nms_capacity, rw, rh, kpts, scores, desc, n_valid,
                  n_candidates, (hipStream_t)stream, desc_f16, &h->prof);
    return check_launch("xfh_detect_sparse");
}

size_t xfh_dense_workspace_bytes(int B, int hc, int wc, int k) {
    if (B <= 0 || hc <= 0 || wc <= 0 || k <= 0) return 0;
    DenseWs o;
    return carve_dense(nullptr, B, hc, wc, k, o);
}

int xfh_extract_dense(xfh_handle h, const float* reliab, const float* feats, int B, int hc, int wc, int k, float rw,
                      float rh, float scale_div, float* kpts, float* desc, int32_t* cell_index, void* workspace,
                      size_t workspace_bytes, xfh_stream stream) {
    if (!h || !reliab || !feats || !kpts || !desc) return fail(XFH_ERR_ARG, "xfh_extract_dense: NULL argument");
    if (B <= 0 || hc <= 0 || wc <= 0 || B > 65535) return fail(XFH_ERR_ARG, "xfh_extract_dense: bad shape");
    if (k <= 0 || k > hc * wc) return fail(XFH_ERR_ARG, "xfh_extract_dense: k %d outside 1..h*w", k);
    if (!(scale_div > 0.f)) return fail(XFH_ERR_ARG, "xfh_extract_dense: scale_div must be positive");
    DenseWs w;
    const size_t need = carve_dense(workspace, B, hc, wc, k, w);
    int rc = check_ws(workspace, workspace_bytes, need);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    launch_topk_desc(reliab, B, hc * wc, k, w.keys, w.skeys, w.nsel, st);
    launch_dense_gather(feats, w.skeys, B, hc, wc, k, rw, rh, scale_div, kpts, desc, cell_index, st);
    return check_launch("xfh_extract_dense");
}

size_t xfh_match_workspace_bytes(int P, int N1, int N2) {
    if (P <= 0 || N1 <= 0 || N2 <= 0) return 0;
    MatchWs o;
    return carve_match(nullptr, P, N1, N2, o);
}

int xfh_match_mnn(xfh_handle h, const float* d1, size_t pair_stride1, const float* d2, size_t pair_stride2,
                  const uint16_t* d1_f16, const uint16_t* d2_f16,
                  const int32_t* n1, const int32_t* n2, int n_stride, int n_offset2, int P, int N1, int N2,
                  float min_cossim, int64_t* idx0, int64_t* idx1, int32_t* n_matches, void* workspace,
                  size_t workspace_bytes, xfh_stream stream) {
    if (!d1 || !d2 || !idx0 || !idx1 || !n_matches) return fail(XFH_ERR_ARG, "xfh_match_mnn: NULL argument");
    if (P <= 0 || N1 <= 0 || N2 <= 0 || P > 65535) return fail(XFH_ERR_ARG, "xfh_match_mnn: bad shape");
    if ((long)P * ((N1 + 1023) / 1024) > 0x7fffffffL / 1024) return fail(XFH_ERR_ARG, "xfh_match_mnn: P * N1 too large");
    if ((pair_stride1 & 3) || (pair_stride2 & 3)) return fail(XFH_ERR_ARG, "xfh_match_mnn: pair strides must be multiples of 4 floats");
    if ((d1_f16 == nullptr) != (d2_f16 == nullptr)) return fail(XFH_ERR_ARG, "xfh_match_mnn: pass both fp16 copies or neither");
    if (d1_f16 && ((pair_stride1 & 7) || (pair_stride2 & 7))) return fail(XFH_ERR_ARG, "xfh_match_mnn: fp16 copies need pair strides that are multiples of 8");
    MatchWs w;
    const size_t need = carve_match(workspace, P, N1, N2, w);
    int rc = check_ws(workspace, workspace_bytes, need);
    if (rc) return rc;
    launch_match(w, d1, pair_stride1, d2, pair_stride2, n1, n2, n_stride, n_offset2, P, N1, N2, min_cossim, idx0, idx1,
                 n_matches, (hipStream_t)stream, h ? &h->prof : nullptr, d1_f16, d2_f16, h ? h->opt.match_exact != 0 : false, h ? h->opt.match_sweep : 0);
    return check_launch("xfh_match_mnn");
}

size_t xfh_refine_workspace_bytes(int P, int N) {
    if (P <= 0 || N <= 0) return 0;
    RefineWs o;
    return carve_refine(nullptr, P, N, o);
}

// the fine_matcher's five layers (modules/model.py:97-111).  Option fx bits 1 | 2048 (default) and every layer with its fragments: the fp16-pair kernels, the activations
// between them in the split form (linear_fx_body.hpp; the same workspace bytes); otherwise the f32-MFMA kernels.  first: the 128-wide input (fp32 rows or the gather of the
// two descriptors); y: (M, 64) fp32
static int fine_chain(xfh_handle h, LinLoader loader, const LinSrc& first, int M, const int32_t* m_dev, float* actA, float* actB, float* y, hipStream_t st) {
    const LinW* f = h->nw.fine;
    bool fx = (h->opt.fx & XFH_FX_FINE) != 0;
    for (int li = 0; li < 5; ++li) fx = fx && f[li].w_fx && (li == 0 || li == 4 || f[li].n_pad % 128 == 0);
    float* bufs[2] = {actA, actB};
    int bad = 0;
    LinSrc r{};
    r.ldx = 512;
    for (int li = 0; li < 5; ++li) {
        const int K = li ? 512 : 128;
        const bool relu = li < 4;
        float* dst = li < 4 ? bufs[li & 1] : y;
        const int ldy = li < 4 ? 512 : 64;
        const LinSrc& src = li ? r : first;
        const LinLoader ld = li ? LOAD_ROWMAJOR : loader;
        if (fx) bad |= launch_linear_fx(f[li].w_fx, f[li].bias, K, f[li].n, f[li].n_pad, relu, ld, src, M, m_dev, dst, ldy, st, h->status, li > 0, li < 4);
        else bad |= launch_linear_mfma(f[li].w_kn, f[li].bias, K, f[li].n, f[li].n_pad, relu, ld, src, M, m_dev, dst, ldy, st);
        r.x = dst;
    }
    return bad;
}

int xfh_refine_matches(xfh_handle h, const float* desc0, const float* desc1, const float* kp0, const float* kp1,
                       const float* scale0, const int64_t* idx0, const int64_t* idx1, const int32_t* n_matches, int P,
                       int N, float fine_conf, float* out, int32_t* n_out, void* workspace, size_t workspace_bytes,
                       xfh_stream stream) {
    if (!h || !desc0 || !desc1 || !kp0 || !kp1 || !scale0 || !idx0 || !idx1 || !n_matches || !out || !n_out)
        return fail(XFH_ERR_ARG, "xfh_refine_matches: NULL argument");
    if (P <= 0 || N <= 0 || P > 65535 || (long)P * N > 0x7fffffffL / 512) return fail(XFH_ERR_ARG, "xfh_refine_matches: bad shape");
    RefineWs w;
    const size_t need = carve_refine(workspace, P, N, w);
    int rc = check_ws(workspace, workspace_bytes, need);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int M = P * N;
    launch_refine_rowmap(n_matches, P, N, w.offs, w.rowmap, w.total, st);
    LinSrc g{};
    g.x = desc0; g.x2 = desc1; g.idx0 = idx0; g.idx1 = idx1; g.rowmap = w.rowmap; g.N = N;
    // (the last layer's (M, 64) lands in actA, as before: layer 3 wrote actB)
    const int bad = fine_chain(h, LOAD_GATHER2, g, M, w.total, w.actA, w.actB, w.actA, st);
    if (bad) return fail(XFH_ERR_UNSUPPORTED, "xfh_refine_matches: missing linear kernel instantiation");
    launch_refine_finish(w.actA, w.rowmap, w.offs, w.total, kp0, kp1, scale0, idx0, idx1, P, N, fine_conf, out, n_out,
                         w.rows, w.keep, st);
    return check_launch("xfh_refine_matches");
}

size_t xfh_homography_workspace_bytes(int P, int max_iters) {
    if (P <= 0 || max_iters <= 0) return 0;
    return xfh::homography_workspace_bytes(P, max_iters);
}

static int find_homography_impl(const char* who, const float* pts0, const float* pts1, const int64_t* idx0, const int64_t* idx1, int kcap,
                                const int32_t* counts, int n_const, int P, int cap, double ransac_thr, int max_iters, double confidence,
                                uint64_t seed, double* H, uint8_t* mask, int32_t* info, void* workspace, size_t workspace_bytes, xfh_stream stream) {
    if (!pts0 || !pts1 || !H || !mask || !info) return fail(XFH_ERR_ARG, "%s: NULL argument", who);
    if (P <= 0 || P > 65535 || cap <= 0 || cap > (1 << 24) || kcap <= 0 || (!counts && (n_const < 0 || n_const > cap)))
        return fail(XFH_ERR_ARG, "%s: bad shape (P %d, cap %d, n %d)", who, P, cap, n_const);
    if (!(ransac_thr > 0.0) || !(confidence > 0.0 && confidence < 1.0)) return fail(XFH_ERR_ARG, "%s: threshold %g / confidence %g", who, ransac_thr, confidence);
    if (max_iters < 1 || max_iters > 4096) return fail(XFH_ERR_UNSUPPORTED, "%s: max_iters %d outside [1, 4096]", who, max_iters);
    int rc = check_ws(workspace, workspace_bytes, xfh::homography_workspace_bytes(P, max_iters));
    if (rc) return rc;
    if (launch_find_homography(pts0, pts1, idx0, idx1, kcap, counts, n_const, P, cap, ransac_thr, max_iters, confidence, seed, H, mask, info, workspace, (hipStream_t)stream))
        return fail(XFH_ERR_UNSUPPORTED, "%s: unsupported configuration", who);
    return check_launch(who);
}

int xfh_find_homography(const float* pts0, const float* pts1, const int32_t* counts, int n_const, int P, int cap, double ransac_thr,
                        int max_iters, double confidence, uint64_t seed, double* H, uint8_t* mask, int32_t* info, void* workspace,
                        size_t workspace_bytes, xfh_stream stream) {
    return find_homography_impl("xfh_find_homography", pts0, pts1, nullptr, nullptr, cap, counts, n_const, P, cap, ransac_thr, max_iters, confidence, seed,
                                H, mask, info, workspace, workspace_bytes, stream);
}

int xfh_find_homography_matches(const float* kpts0, const float* kpts1, int kpt_cap, const int64_t* idx0, const int64_t* idx1,
                                const int32_t* n_matches, int P, int cap, double ransac_thr, int max_iters, double confidence, uint64_t seed,
                                double* H, uint8_t* mask, int32_t* info, void* workspace, size_t workspace_bytes, xfh_stream stream) {
    if (!idx0 || !idx1 || !n_matches) return fail(XFH_ERR_ARG, "xfh_find_homography_matches: NULL argument");
    return find_homography_impl("xfh_find_homography_matches", kpts0, kpts1, idx0, idx1, kpt_cap, n_matches, 0, P, cap, ransac_thr, max_iters, confidence, seed,
                                H, mask, info, workspace, workspace_bytes, stream);
}

int xfh_homography_tables(double ransac_thr, uint32_t* score_table, double* weight_table, xfh_stream stream) {
    if (!score_table || !weight_table || !(ransac_thr > 0.0)) return fail(XFH_ERR_ARG, "xfh_homography_tables: bad argument");
    launch_homography_tables(ransac_thr, score_table, weight_table, (hipStream_t)stream);
    return check_launch("xfh_homography_tables");
}

int xfh_kpts_heatmap(const float* logits, int B, int hc, int wc, float* heat, xfh_stream stream) {
    if (!logits || !heat || B <= 0 || hc <= 0 || wc <= 0) return fail(XFH_ERR_ARG, "xfh_kpts_heatmap: bad argument");
    launch_softmax_heat(logits, B, hc, wc, heat, (hipStream_t)stream);
    return check_launch("xfh_kpts_heatmap");
}

int xfh_sample_sparse(const float* x, const float* pos, int B, int C, int Hm, int Wm, int N, int H, int W, int mode, float* out,
                      xfh_stream stream) {
    if (!x || !pos || !out) return fail(XFH_ERR_ARG, "xfh_sample_sparse: NULL argument");
    if (B <= 0 || C <= 0 || Hm <= 0 || Wm <= 0 || N < 0 || H <= 1 || W <= 1) return fail(XFH_ERR_ARG, "xfh_sample_sparse: bad shape");
    if (mode < XFH_SAMPLE_NEAREST || mode > XFH_SAMPLE_BICUBIC) return fail(XFH_ERR_ARG, "xfh_sample_sparse: mode %d", mode);
    if ((double)B * N * C >= 4294967296.0 * 256) return fail(XFH_ERR_UNSUPPORTED, "xfh_sample_sparse: too many samples");
    if (N == 0) return XFH_OK;
    launch_sample_sparse(x, pos, B, C, Hm, Wm, N, H, W, mode, out, (hipStream_t)stream);
    return check_launch("xfh_sample_sparse");
}

int xfh_nms(xfh_handle h, const float* heat, int B, int H, int W, float threshold, int kernel_size, int capacity, int64_t* xy,
            int32_t* n_candidates, void* workspace, size_t workspace_bytes, xfh_stream stream) {
    (void)h;
    if (!heat || !xy || !n_candidates) return fail(XFH_ERR_ARG, "xfh_nms: NULL argument");
    if (kernel_size < 1 || !(kernel_size & 1) || kernel_size > 255) return fail(XFH_ERR_ARG, "xfh_nms: kernel_size %d must be odd, 1..255", kernel_size);
    if (B <= 0 || H <= 0 || W <= 0 || H >= 65536 || W >= 65536 || B > 65535) return fail(XFH_ERR_ARG, "xfh_nms: bad shape");
    if (capacity <= 0 || (long)capacity > (long)H * W) return fail(XFH_ERR_ARG, "xfh_nms: capacity outside 1..H*W");
    DetectWs w;
    const size_t need = carve_detect(workspace, B, H, W, 1, capacity, w);
    int rc = check_ws(workspace, workspace_bytes, need);
    if (rc) return rc;
    launch_nms_only(w, heat, B, H, W, threshold, kernel_size, capacity, xy, n_candidates, (hipStream_t)stream);
    return check_launch("xfh_nms");
}

int xfh_fine_matcher(xfh_handle h, const float* x, int n, float* out, void* workspace, size_t workspace_bytes,
                     xfh_stream stream) {
    if (!h || !x || !out || n <= 0) return fail(XFH_ERR_ARG, "xfh_fine_matcher: bad argument");
    RefineWs w;
    const size_t need = carve_refine(workspace, 1, n, w);
    int rc = check_ws(workspace, workspace_bytes, need);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    LinSrc r{};
    r.ldx = 128; r.x = x;
    const int bad = fine_chain(h, LOAD_ROWMAJOR, r, n, nullptr, w.actA, w.actB, out, st);
    if (bad) return fail(XFH_ERR_UNSUPPORTED, "xfh_fine_matcher: missing linear kernel instantiation");
    return check_launch("xfh_fine_matcher");
}

int xfh_debug_match_occupancy(void) { return xfh::match_debug_occupancy(); }
int xfh_debug_cold_start(int enable) { xfh::g_debug_cold = enable ? 1 : 0; return XFH_OK; }
int xfh_debug_block1(xfh_handle h, const float* gray, const float* coef, int B, int H, int W, float* x1, xfh_stream stream) {
    if (!h || !gray || !coef || !x1) return fail(XFH_ERR_ARG, "xfh_debug_block1: NULL argument");
    if (B <= 0 || H <= 0 || W <= 0 || (H & 3) || (W & 3)) return fail(XFH_ERR_ARG, "xfh_debug_block1: bad shape (%d,%d,%d)", B, H, W);
    launch_block1_fused(h->nw, gray, coef, B, H, W, x1, (hipStream_t)stream, h->opt.block1, h->status);
    return check_launch("xfh_debug_block1");
}

// option -> its slot and the values it takes (include/xfeat_hip.h: xfh_set_option)
static int* option_slot(xfh_handle h, const char* key) {
    struct { const char* k; int Options::*m; } tab[] = {{"match_exact", &Options::match_exact}, {"match_sweep", &Options::match_sweep}, {"block1", &Options::block1}, {"fx", &Options::fx}, {"resize2", &Options::resize2}};
    for (auto& t : tab)
        if (!strcmp(t.k, key)) return &(h->opt.*(t.m));
    return nullptr;
}
int xfh_set_option(xfh_handle h, const char* key, int value) {
    if (!h || !key) return fail(XFH_ERR_ARG, "xfh_set_option: NULL argument");
    int* slot = option_slot(h, key);
    if (!slot) return fail(XFH_ERR_ARG, "xfh_set_option: unknown option '%s' (match_exact, match_sweep, block1, fx, resize2)", key);
    bool ok;
    if (!strcmp(key, "block1")) ok = value == 5 || value == 7;
    else if (!strcmp(key, "fx")) ok = value >= 0 && (value & ~XFH_FX_ALL) == 0;
    else if (!strcmp(key, "match_sweep")) ok = value >= 0 && value <= 2;
    else ok = value == 0 || value == 1;
    if (!ok) return fail(XFH_ERR_ARG, "xfh_set_option: %s = %d is not a value of this option", key, value);
    *slot = value;
    return XFH_OK;
}
int xfh_set_status_buffer(xfh_handle h, int32_t* device_word) {
    if (!h) return fail(XFH_ERR_ARG, "xfh_set_status_buffer: NULL handle");
    h->status = reinterpret_cast<int*>(device_word);
    return XFH_OK;
}
int xfh_get_option(xfh_handle h, const char* key, int* value) {
    if (!h || !key || !value) return fail(XFH_ERR_ARG, "xfh_get_option: NULL argument");
    const int* slot = option_slot(h, key);
    if (!slot) return fail(XFH_ERR_ARG, "xfh_get_option: unknown option '%s'", key);
    *value = *slot;
    return XFH_OK;
}

int xfh_debug_trace(xfh_handle h, long long* device_buffer) {
    if (!h) return fail(XFH_ERR_ARG, "xfh_debug_trace: NULL handle");
    h->trace = device_buffer;
    g_head_trace = device_buffer ? device_buffer + (1 << 21) + (1 << 16) : nullptr;      // the key-point head's stamps live 2 Mi + 64 Ki entries into the buffer (the conv kernels use the front)
    return XFH_OK;
}

int xfh_profile_select(xfh_handle h, int which) {
    if (!h) return fail(XFH_ERR_ARG, "xfh_profile_select: NULL handle");
    h->prof.which = which;
    h->prof.used = 0;
    h->prof.flops = h->prof.bytes = 0;
    return XFH_OK;
}

int xfh_profile_read_spans(xfh_handle h, int* ids, double* ms, int capacity, int* n_spans) {
    if (!h || !n_spans) return fail(XFH_ERR_ARG, "xfh_profile_read_spans: NULL argument");
    const int n = (int)(h->prof.used / 2);
    for (int i = 0; i < n && i < capacity; ++i) {
        HIP_TRY(hipEventSynchronize(h->prof.ev[2 * i + 1]));
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, h->prof.ev[2 * i], h->prof.ev[2 * i + 1]));
        if (ids) ids[i] = h->prof.ids[i];
        if (ms) ms[i] = t;
    }
    *n_spans = n;
    h->prof.used = 0;
    h->prof.flops = h->prof.bytes = 0;
    return XFH_OK;
}

int xfh_profile_read(xfh_handle h, int* n_launches, double* total_ms, double* total_flops, double* total_bytes) {
    if (!h) return fail(XFH_ERR_ARG, "xfh_profile_read: NULL handle");
    double ms = 0;
    for (size_t i = 0; i + 1 < h->prof.used; i += 2) {
        HIP_TRY(hipEventSynchronize(h->prof.ev[i + 1]));
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, h->prof.ev[i], h->prof.ev[i + 1]));
        ms += t;
    }
    if (n_launches) *n_launches = (int)(h->prof.used / 2);
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = h->prof.flops;
    if (total_bytes) *total_bytes = h->prof.bytes;
    h->prof.used = 0;
    h->prof.flops = h->prof.bytes = 0;
    return XFH_OK;
}

}  // extern "C"
