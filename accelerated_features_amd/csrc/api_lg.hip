// C ABI of the LighterGlue matcher (include/xfeat_hip.h, section "LighterGlue"): weight packing + the per-pair
// schedule of kernels.  Replaces kornia.feature.lightglue.LightGlue.forward as configured by
// modules/lighterglue.py:12-27 and called from modules/xfeat.py:131-162.
#include "../../include/xfeat_hip.h"
#include "kernels.hpp"
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace xfh;

int xfh_set_error(int code, const char* fmt, ...);      // api.hip

namespace {
constexpr int LG_LAYERS = 6, LG_D = 96, LG_IN = 64;
struct LgLin { const float* w; const float* b; int k, n, npad; };      // w: [k][npad] (K-major, zero padded), b: [npad]
struct LgFfn { LgLin l0, l3; const float* gamma; const float* beta; };
struct LgLayer {
    LgLin wqkv, out_proj;   // self:  Wqkv rows re-ordered to [q | k | v]
    LgFfn self_ffn;
    LgLin to_qk, to_v, to_out;   // cross: to_qk carries the 96^-1/4 scale
    LgFfn cross_ffn;
    const float* match_w; const float* match_b;      // log_assignment[i].matchability
};
}  // namespace

struct xfh_lg_context {
    int device;
    float* blob;
    LgLin input_proj, final_proj;     // final_proj = log_assignment[5].final_proj / 96^(1/4)
    const float* wr;                  // posenc.Wr (48,2)
    LgLayer layer[LG_LAYERS];
};

// canonical array order = oracle/lighterglue_oracle.py::state_dict_keys() = kornia module order
static void lg_shapes(std::vector<std::pair<int, int>>& sh) {      // (rows, cols); 1-D arrays as (n, 1)
    const int d = LG_D;
    sh.push_back({d, LG_IN}); sh.push_back({d, 1}); sh.push_back({d / 2, 2});
    for (int i = 0; i < LG_LAYERS; ++i) {
        sh.push_back({3 * d, d}); sh.push_back({3 * d, 1}); sh.push_back({d, d}); sh.push_back({d, 1});
        sh.push_back({2 * d, 2 * d}); sh.push_back({2 * d, 1}); sh.push_back({2 * d, 1}); sh.push_back({2 * d, 1}); sh.push_back({d, 2 * d}); sh.push_back({d, 1});
        sh.push_back({d, d}); sh.push_back({d, 1}); sh.push_back({d, d}); sh.push_back({d, 1}); sh.push_back({d, d}); sh.push_back({d, 1});
        sh.push_back({2 * d, 2 * d}); sh.push_back({2 * d, 1}); sh.push_back({2 * d, 1}); sh.push_back({2 * d, 1}); sh.push_back({d, 2 * d}); sh.push_back({d, 1});
    }
    for (int i = 0; i < LG_LAYERS; ++i) { sh.push_back({1, d}); sh.push_back({1, 1}); sh.push_back({d, d}); sh.push_back({d, 1}); }
    for (int i = 0; i < LG_LAYERS - 1; ++i) { sh.push_back({1, d}); sh.push_back({1, 1}); }
}

int xfh_lg_num_weight_arrays(void) {
    std::vector<std::pair<int, int>> sh; lg_shapes(sh);
    return (int)sh.size();
}
size_t xfh_lg_weight_array_floats(int i) {
    std::vector<std::pair<int, int>> sh; lg_shapes(sh);
    if (i < 0 || i >= (int)sh.size()) return 0;
    return (size_t)sh[i].first * sh[i].second;
}

int xfh_lg_create(const float* const* host_arrays, int n_arrays, int device, xfh_lg_handle* out) {
    if (!host_arrays || !out) return xfh_set_error(XFH_ERR_ARG, "xfh_lg_create: NULL argument");
    std::vector<std::pair<int, int>> sh; lg_shapes(sh);
    if (n_arrays != (int)sh.size()) return xfh_set_error(XFH_ERR_WEIGHTS, "xfh_lg_create: expected %d weight arrays, got %d", (int)sh.size(), n_arrays);
    for (int i = 0; i < n_arrays; ++i)
        if (!host_arrays[i]) return xfh_set_error(XFH_ERR_WEIGHTS, "xfh_lg_create: weight array %d is NULL", i);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return xfh_set_error(XFH_ERR_DEVICE, "xfh_lg_create: no HIP device visible");
    if (device < 0 || device >= ndev) return xfh_set_error(XFH_ERR_DEVICE, "xfh_lg_create: device %d out of range", device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return xfh_set_error(XFH_ERR_DEVICE, "xfh_lg_create: device %d is not gfx950", device);
    if (hipSetDevice(device) != hipSuccess) return xfh_set_error(XFH_ERR_HIP, "xfh_lg_create: hipSetDevice failed");

    std::vector<float> blob;
    auto reserve = [&](size_t n) { size_t o = (blob.size() + 63) / 64 * 64; blob.resize(o + n, 0.f); return o; };
    struct LinOff { size_t w, b; int k, n, npad; };
    int ai = 0;
    // Linear (n, k) + bias -> K-major [k][npad]; `perm` maps output feature -> source row; `scale` folds a constant
    auto pack_lin = [&](int n, int k, double scale, bool qkv_perm) {
        const float* w = host_arrays[ai++];
        const float* b = host_arrays[ai++];
        LinOff o; o.k = k; o.n = n; o.npad = (n + 63) / 64 * 64;
        o.w = reserve((size_t)k * o.npad); o.b = reserve(o.npad);
        for (int j = 0; j < n; ++j) {
            const int src = qkv_perm ? 3 * (j % LG_D) + j / LG_D : j;     // [q|k|v] <- interleaved (c,t) rows of kornia's Wqkv
            for (int c = 0; c < k; ++c) blob[o.w + (size_t)c * o.npad + j] = (float)((double)w[(size_t)src * k + c] * scale);
            blob[o.b + j] = (float)((double)b[src] * scale);
        }
        return o;
    };
    auto pack_vec = [&](int n) { const float* v = host_arrays[ai++]; size_t o = reserve(n); memcpy(&blob[o], v, n * sizeof(float)); return o; };
    const LinOff o_in = pack_lin(LG_D, LG_IN, 1.0, false);
    const size_t o_wr = pack_vec(LG_D);     // (48,2) row-major
    struct LayerOff { LinOff wqkv, outp, s0, s3, qk, v, to, c0, c3; size_t sg, sb, cg, cb, mw, mb; LinOff fp; } lo[LG_LAYERS];
    const double qk_scale = std::pow((double)LG_D, -0.25);
    for (int i = 0; i < LG_LAYERS; ++i) {
        lo[i].wqkv = pack_lin(3 * LG_D, LG_D, 1.0, true);
        lo[i].outp = pack_lin(LG_D, LG_D, 1.0, false);
        lo[i].s0 = pack_lin(2 * LG_D, 2 * LG_D, 1.0, false);
        lo[i].sg = pack_vec(2 * LG_D); lo[i].sb = pack_vec(2 * LG_D);
        lo[i].s3 = pack_lin(LG_D, 2 * LG_D, 1.0, false);
        lo[i].qk = pack_lin(LG_D, LG_D, qk_scale, false);
        lo[i].v = pack_lin(LG_D, LG_D, 1.0, false);
        lo[i].to = pack_lin(LG_D, LG_D, 1.0, false);
        lo[i].c0 = pack_lin(2 * LG_D, 2 * LG_D, 1.0, false);
        lo[i].cg = pack_vec(2 * LG_D); lo[i].cb = pack_vec(2 * LG_D);
        lo[i].c3 = pack_lin(LG_D, 2 * LG_D, 1.0, false);
    }
    for (int i = 0; i < LG_LAYERS; ++i) {
        lo[i].mw = pack_vec(LG_D); lo[i].mb = pack_vec(1);
        lo[i].fp = pack_lin(LG_D, LG_D, qk_scale, false);       // final_proj / d^(1/4)
    }
    // token_confidence heads: unused (depth_confidence = -1 disables early stopping, and with it the confidences)
    ai += 2 * (LG_LAYERS - 1);

    xfh_lg_context* ctx = new xfh_lg_context();
    ctx->device = device; ctx->blob = nullptr;
    hipError_t e = hipMalloc((void**)&ctx->blob, blob.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(ctx->blob, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (ctx->blob) (void)hipFree(ctx->blob);
        delete ctx;
        return xfh_set_error(XFH_ERR_HIP, "xfh_lg_create: weight upload failed: %s", hipGetErrorString(e));
    }
    auto lin = [&](const LinOff& o) { LgLin l; l.w = ctx->blob + o.w; l.b = ctx->blob + o.b; l.k = o.k; l.n = o.n; l.npad = o.npad; return l; };
    ctx->input_proj = lin(o_in);
    ctx->wr = ctx->blob + o_wr;
    for (int i = 0; i < LG_LAYERS; ++i) {
        LgLayer& L = ctx->layer[i];
        L.wqkv = lin(lo[i].wqkv); L.out_proj = lin(lo[i].outp);
        L.self_ffn.l0 = lin(lo[i].s0); L.self_ffn.l3 = lin(lo[i].s3); L.self_ffn.gamma = ctx->blob + lo[i].sg; L.self_ffn.beta = ctx->blob + lo[i].sb;
        L.to_qk = lin(lo[i].qk); L.to_v = lin(lo[i].v); L.to_out = lin(lo[i].to);
        L.cross_ffn.l0 = lin(lo[i].c0); L.cross_ffn.l3 = lin(lo[i].c3); L.cross_ffn.gamma = ctx->blob + lo[i].cg; L.cross_ffn.beta = ctx->blob + lo[i].cb;
        L.match_w = ctx->blob + lo[i].mw; L.match_b = ctx->blob + lo[i].mb;
    }
    ctx->final_proj = lin(lo[LG_LAYERS - 1].fp);
    *out = ctx;
    return XFH_OK;
}

void xfh_lg_destroy(xfh_lg_handle h) {
    if (!h) return;
    if (h->blob) (void)hipFree(h->blob);
    delete h;
}

namespace {
struct LgSet {
    float *xa, *xb, *csa, *csb, *sna, *snb;     // ping-pong under pruning: x (N,192) = [descriptor | message]
    int32_t *inda, *indb, *map, *na, *nb;
    float *qkv, *hid, *tmp, *z, *md;
};
struct LgWs {
    LgSet s[2];
    float *md1t, *sim, *rlse, *clse, *best0, *zeros;
    int32_t *m0, *m1;
    int n1pad;
};
struct Carver {
    char* base; size_t off = 0;
    explicit Carver(void* p) : base((char*)p) {}
    template <typename T> T* take(size_t n) { off = (off + 255) / 256 * 256; T* r = base ? (T*)(base + off) : nullptr; off += n * sizeof(T); return r; }
};
size_t carve_lg(void* ws, int N0, int N1, LgWs& w) {
    Carver c(ws);
    const int N[2] = {N0, N1};
    for (int s = 0; s < 2; ++s) {
        const size_t n = N[s];
        LgSet& S = w.s[s];
        S.xa = c.take<float>(n * 192); S.xb = c.take<float>(n * 192);
        S.csa = c.take<float>(n * 96); S.csb = c.take<float>(n * 96); S.sna = c.take<float>(n * 96); S.snb = c.take<float>(n * 96);
        S.inda = c.take<int32_t>(n); S.indb = c.take<int32_t>(n); S.map = c.take<int32_t>(n); S.na = c.take<int32_t>(1); S.nb = c.take<int32_t>(1);
        S.qkv = c.take<float>(n * 320); S.hid = c.take<float>(n * 192); S.tmp = c.take<float>(n * 128); S.z = c.take<float>(n); S.md = c.take<float>(n * 128);
    }
    w.n1pad = (N1 + 63) / 64 * 64;
    w.md1t = c.take<float>((size_t)96 * w.n1pad);
    w.sim = c.take<float>((size_t)N0 * w.n1pad);
    w.rlse = c.take<float>(N0); w.clse = c.take<float>(w.n1pad); w.best0 = c.take<float>(N0); w.zeros = c.take<float>(w.n1pad);
    w.m0 = c.take<int32_t>(N0); w.m1 = c.take<int32_t>(w.n1pad);
    return (c.off + 255) / 256 * 256;
}
__global__ void lg_init_kernel(int32_t* ind, int n, int32_t* count, float* zeros, int nz) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < n) ind[g] = g;
    if (g < nz && zeros) zeros[g] = 0.f;
    if (g == 0) *count = n;
}
int lin(const LgLin& L, const float* x, int ldx, int cap, const int32_t* n_dev, float* y, int ldy, hipStream_t st) {
    LinSrc s{};
    s.x = x; s.ldx = ldx;
    return launch_linear_mfma(L.w, L.b, L.k, L.n, L.npad, false, LOAD_ROWMAJOR, s, cap, n_dev, y, ldy, st);
}
}  // namespace

size_t xfh_lg_workspace_bytes(int N0, int N1) {
    if (N0 <= 0 || N1 <= 0) return 0;
    LgWs w;
    return carve_lg(nullptr, N0, N1, w);
}

int xfh_lg_match(xfh_lg_handle h, const float* kpts0, const float* desc0, int N0, float W0, float H0, const float* kpts1, const float* desc1,
                 int N1, float W1, float H1, float min_conf, int prune_min_kpts, int64_t* matches, float* scores, int32_t* n_matches, void* workspace,
                 size_t workspace_bytes, xfh_stream stream) {
    if (!h || !kpts0 || !desc0 || !kpts1 || !desc1 || !matches || !scores || !n_matches) return xfh_set_error(XFH_ERR_ARG, "xfh_lg_match: NULL argument");
    if (N0 <= 0 || N1 <= 0 || N0 > 16384 || N1 > 16384) return xfh_set_error(XFH_ERR_ARG, "xfh_lg_match: key-point counts must be in 1..16384");
    LgWs w;
    const size_t need = carve_lg(workspace, N0, N1, w);
    if (!workspace || workspace_bytes < need || ((size_t)workspace & 255)) return xfh_set_error(XFH_ERR_WORKSPACE, "xfh_lg_match: workspace too small or misaligned (%zu < %zu)", workspace_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    const int N[2] = {N0, N1};
    const float* kp[2] = {kpts0, kpts1};
    const float* de[2] = {desc0, desc1};
    const float Wd[2] = {W0, W1}, Hd[2] = {H0, H1};
    int bad = 0;
    // current (x, cos, sin, ind, n) of each set
    float *x[2], *cs[2], *sn[2]; int32_t *ind[2], *nn[2];
    for (int s = 0; s < 2; ++s) {
        LgSet& S = w.s[s];
        x[s] = S.xa; cs[s] = S.csa; sn[s] = S.sna; ind[s] = S.inda; nn[s] = S.na;
        lg_init_kernel<<<ceil_div(max(N[s], s ? w.n1pad : 1), 256), 256, 0, st>>>(ind[s], N[s], nn[s], s ? w.zeros : nullptr, s ? w.n1pad : 0);
        launch_lg_encode(kp[s], N[s], Wd[s], Hd[s], h->wr, cs[s], sn[s], st);
        bad |= lin(h->input_proj, de[s], LG_IN, N[s], nullptr, x[s], 192, st);
    }
    auto ffn = [&](const LgFfn& F, int s) {
        LgSet& S = w.s[s];
        bad |= lin(F.l0, x[s], 192, N[s], nn[s], S.hid, 192, st);
        launch_lg_ln_gelu(S.hid, 192, nn[s], N[s], F.gamma, F.beta, st);
        bad |= lin(F.l3, S.hid, 192, N[s], nn[s], S.tmp, 128, st);
        launch_lg_add(x[s], 192, S.tmp, 128, nn[s], N[s], st);
    };
    const float self_scale = 1.0f / std::sqrt((float)LG_D);
    for (int i = 0; i < LG_LAYERS; ++i) {
        const LgLayer& L = h->layer[i];
        for (int s = 0; s < 2; ++s) {          // self block
            LgSet& S = w.s[s];
            bad |= lin(L.wqkv, x[s], 192, N[s], nn[s], S.qkv, 320, st);
            launch_lg_rotary(S.qkv, 320, nn[s], N[s], cs[s], sn[s], st);
            launch_lg_attention(S.qkv, 320, S.qkv + 96, 320, S.qkv + 192, 320, S.tmp, 128, nn[s], nn[s], N[s], N[s], self_scale, st);
            bad |= lin(L.out_proj, S.tmp, 128, N[s], nn[s], x[s] + 96, 192, st);
            ffn(L.self_ffn, s);
        }
        for (int s = 0; s < 2; ++s) {          // cross block: projections of both sets first
            LgSet& S = w.s[s];
            bad |= lin(L.to_qk, x[s], 192, N[s], nn[s], S.qkv, 320, st);
            bad |= lin(L.to_v, x[s], 192, N[s], nn[s], S.qkv + 96, 320, st);
        }
        for (int s = 0; s < 2; ++s) {
            LgSet &S = w.s[s], &T = w.s[s ^ 1];
            launch_lg_attention(S.qkv, 320, T.qkv, 320, T.qkv + 96, 320, S.tmp, 128, nn[s], nn[s ^ 1], N[s], N[s ^ 1], 1.0f, st);
        }
        for (int s = 0; s < 2; ++s) {
            bad |= lin(L.to_out, w.s[s].tmp, 128, N[s], nn[s], x[s] + 96, 192, st);
            ffn(L.cross_ffn, s);
        }
        if (i == LG_LAYERS - 1 || prune_min_kpts >= (1 << 30)) continue;
        for (int s = 0; s < 2; ++s) {          // width pruning: matchability > 1 - width_confidence (0.95)
            LgSet& S = w.s[s];
            launch_lg_dot(x[s], 192, nn[s], N[s], L.match_w, L.match_b, S.z, st);
            const bool a = x[s] == S.xa;
            float *xo = a ? S.xb : S.xa, *cso = a ? S.csb : S.csa, *sno = a ? S.snb : S.sna;
            int32_t *indo = a ? S.indb : S.inda, *no = a ? S.nb : S.na;
            launch_lg_prune(S.z, 0.05f, prune_min_kpts, nn[s], N[s], S.map, no, x[s], 192, xo, cs[s], cso, sn[s], sno, ind[s], indo, st);
            x[s] = xo; cs[s] = cso; sn[s] = sno; ind[s] = indo; nn[s] = no;
        }
    }
    // assignment of the last layer
    const LgLayer& L = h->layer[LG_LAYERS - 1];
    for (int s = 0; s < 2; ++s) {
        bad |= lin(h->final_proj, x[s], 192, N[s], nn[s], w.s[s].md, 128, st);
        launch_lg_dot(x[s], 192, nn[s], N[s], L.match_w, L.match_b, w.s[s].z, st);
    }
    launch_lg_transpose(w.s[1].md, 128, nn[1], N1, w.md1t, w.n1pad, st);
    {
        LinSrc src{};
        src.x = w.s[0].md; src.ldx = 128;
        bad |= launch_linear_mfma(w.md1t, w.zeros, LG_D, w.n1pad, w.n1pad, false, LOAD_ROWMAJOR, src, N0, nn[0], w.sim, w.n1pad, st);
    }
    launch_lg_assign(w.sim, w.n1pad, nn[0], N0, nn[1], N1, w.s[0].z, w.s[1].z, w.rlse, w.clse, w.m0, w.m1, w.best0, ind[0], ind[1], min_conf,
                     matches, scores, n_matches, st);
    if (bad) return xfh_set_error(XFH_ERR_UNSUPPORTED, "xfh_lg_match: missing linear kernel instantiation");
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xfh_set_error(XFH_ERR_HIP, "xfh_lg_match: %s", hipGetErrorString(e));
    return XFH_OK;
}
