// C ABI of the LighterGlue matcher (include/xfeat_hip.h, section "LighterGlue"): weight packing + the per-pair
// schedule of kernels.  Replaces kornia.feature.lightglue.LightGlue.forward as configured by
// modules/lighterglue.py:12-27 and called from modules/xfeat.py:131-162.
#include "../../include/xfeat_hip.h"
#include "kernels.hpp"
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace xfh;

int xfh_set_error(int code, const char* fmt, ...);      // api.hip

namespace {
constexpr int LG_LAYERS = 6, LG_D = 96, LG_IN = 64;
struct LgLin { const float* w; const float* b; int k, n; };      // w: operand order [n/32][2][k/8][32][4], b: [n]
struct LgFfn { LgLin l0, l3; const float* gamma; const float* beta; };
struct LgLayer {
    LgLin wqkv, out_proj;   // self:  Wqkv rows re-ordered to [q | k | v]
    LgFfn self_ffn;
    LgLin to_qkv, to_out;   // cross: rows 0..95 = to_qk (carries the 96^-1/4 scale), rows 96..191 = to_v
    LgFfn cross_ffn;
    const float* match_w; const float* match_b;      // log_assignment[i].matchability
};
}  // namespace

struct xfh_lg_context {
    int device;
    float* blob;
    // bench hook (xfh_lg_profile): HIP events around every attention launch + its algorithmic FLOPs at the capacities
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;
    size_t prof_used = 0;
    double prof_flops = 0;
    LgLin input_proj, final_proj;     // final_proj = log_assignment[5].final_proj / 96^(1/4)
    const float* wr;                  // posenc.Wr (48,2)
    LgLayer layer[LG_LAYERS];
};

// canonical array order = kornia's module order (include/xfeat_hip.h lists it)
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
    struct LinOff { size_t w, b; int k, n; };
    int ai = 0;
    // `parts` consecutive Linear (n_each, k) + bias of host_arrays, stacked along the output dimension, into the operand
    // order of lg_linear_kernel: float4 index ((cb*2 + half)*(k/8) + j)*32 + lane = W[cb*32 + lane][half*k/2 + 4j .. +3].
    // scale[p] folds a constant into part p; qkv_perm re-orders kornia's interleaved (c,t) rows of Wqkv to [q|k|v].
    auto pack_lin = [&](int n_each, int k, int parts, const double* scale, bool qkv_perm) {
        LinOff o; o.k = k; o.n = n_each * parts;
        o.w = reserve((size_t)k * o.n); o.b = reserve(o.n);
        for (int p = 0; p < parts; ++p) {
            const float* w = host_arrays[ai++];
            const float* b = host_arrays[ai++];
            for (int jn = 0; jn < n_each; ++jn) {
                const int src = qkv_perm ? 3 * (jn % LG_D) + jn / LG_D : jn;
                const int out = p * n_each + jn, cb = out / 32, ln = out % 32;
                for (int c = 0; c < k; ++c) {
                    const int half = c / (k / 2), cc = c % (k / 2);
                    const size_t idx = ((((size_t)cb * 2 + half) * (k / 8) + cc / 4) * 32 + ln) * 4 + cc % 4;
                    blob[o.w + idx] = (float)((double)w[(size_t)src * k + c] * scale[p]);
                }
                blob[o.b + out] = (float)((double)b[src] * scale[p]);
            }
        }
        return o;
    };
    auto pack_vec = [&](int n) { const float* v = host_arrays[ai++]; size_t o = reserve(n); memcpy(&blob[o], v, n * sizeof(float)); return o; };
    const double one[2] = {1.0, 1.0};
    const double qk_scale = std::pow((double)LG_D, -0.25);
    const double qkv_scale[2] = {qk_scale, 1.0}, fp_scale[1] = {qk_scale};
    const LinOff o_in = pack_lin(LG_D, LG_IN, 1, one, false);
    const size_t o_wr = pack_vec(LG_D);     // (48,2) row-major
    struct LayerOff { LinOff wqkv, outp, s0, s3, qkv, to, c0, c3; size_t sg, sb, cg, cb, mw, mb; LinOff fp; } lo[LG_LAYERS];
    for (int i = 0; i < LG_LAYERS; ++i) {
        lo[i].wqkv = pack_lin(3 * LG_D, LG_D, 1, one, true);
        lo[i].outp = pack_lin(LG_D, LG_D, 1, one, false);
        lo[i].s0 = pack_lin(2 * LG_D, 2 * LG_D, 1, one, false);
        lo[i].sg = pack_vec(2 * LG_D); lo[i].sb = pack_vec(2 * LG_D);
        lo[i].s3 = pack_lin(LG_D, 2 * LG_D, 1, one, false);
        lo[i].qkv = pack_lin(LG_D, LG_D, 2, qkv_scale, false);      // to_qk then to_v: adjacent in the array order
        lo[i].to = pack_lin(LG_D, LG_D, 1, one, false);
        lo[i].c0 = pack_lin(2 * LG_D, 2 * LG_D, 1, one, false);
        lo[i].cg = pack_vec(2 * LG_D); lo[i].cb = pack_vec(2 * LG_D);
        lo[i].c3 = pack_lin(LG_D, 2 * LG_D, 1, one, false);
    }
    for (int i = 0; i < LG_LAYERS; ++i) {
        lo[i].mw = pack_vec(LG_D); lo[i].mb = pack_vec(1);
        lo[i].fp = pack_lin(LG_D, LG_D, 1, fp_scale, false);       // final_proj / d^(1/4)
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
    auto lin = [&](const LinOff& o) { LgLin l; l.w = ctx->blob + o.w; l.b = ctx->blob + o.b; l.k = o.k; l.n = o.n; return l; };
    ctx->input_proj = lin(o_in);
    ctx->wr = ctx->blob + o_wr;
    for (int i = 0; i < LG_LAYERS; ++i) {
        LgLayer& L = ctx->layer[i];
        L.wqkv = lin(lo[i].wqkv); L.out_proj = lin(lo[i].outp);
        L.self_ffn.l0 = lin(lo[i].s0); L.self_ffn.l3 = lin(lo[i].s3); L.self_ffn.gamma = ctx->blob + lo[i].sg; L.self_ffn.beta = ctx->blob + lo[i].sb;
        L.to_qkv = lin(lo[i].qkv); L.to_out = lin(lo[i].to);
        L.cross_ffn.l0 = lin(lo[i].c0); L.cross_ffn.l3 = lin(lo[i].c3); L.cross_ffn.gamma = ctx->blob + lo[i].cg; L.cross_ffn.beta = ctx->blob + lo[i].cb;
        L.match_w = ctx->blob + lo[i].mw; L.match_b = ctx->blob + lo[i].mb;
    }
    ctx->final_proj = lin(lo[LG_LAYERS - 1].fp);
    *out = ctx;
    return XFH_OK;
}

int xfh_lg_profile(xfh_lg_handle h, int enable) {
    if (!h) return xfh_set_error(XFH_ERR_ARG, "xfh_lg_profile: NULL handle");
    h->prof_on = enable != 0;
    h->prof_used = 0;
    h->prof_flops = 0;
    return XFH_OK;
}

int xfh_lg_profile_read(xfh_lg_handle h, int* n_launches, double* total_ms, double* total_flops) {
    if (!h) return xfh_set_error(XFH_ERR_ARG, "xfh_lg_profile_read: NULL handle");
    double ms = 0;
    for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
        float t = 0;
        if (hipEventSynchronize(h->prof_ev[i + 1]) != hipSuccess || hipEventElapsedTime(&t, h->prof_ev[i], h->prof_ev[i + 1]) != hipSuccess)
            return xfh_set_error(XFH_ERR_HIP, "xfh_lg_profile_read: event query failed");
        ms += t;
    }
    if (n_launches) *n_launches = (int)(h->prof_used / 2);
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = h->prof_flops;
    h->prof_used = 0;
    h->prof_flops = 0;
    return XFH_OK;
}

void xfh_lg_destroy(xfh_lg_handle h) {
    if (!h) return;
    for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
    if (h->blob) (void)hipFree(h->blob);
    delete h;
}

namespace {
struct LgSet {
    float *xa, *xb, *csa, *csb, *sna, *snb;     // ping-pong under pruning: x (N,192) = [descriptor | message]
    int32_t *inda, *indb, *map, *na, *nb;
    float *qkv, *hid, *att, *z, *md;
};
struct LgWs {
    LgSet s[2];
    float *md1t, *sim, *rlse, *clse, *best0, *zeros, *part;
    void* ascratch;
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
        S.qkv = c.take<float>(n * 288); S.hid = c.take<float>(n * 192); S.att = c.take<float>(n * 96); S.z = c.take<float>(n + 64); S.md = c.take<float>(n * 96);
    }
    w.n1pad = (N1 + 63) / 64 * 64;
    w.md1t = c.take<float>((size_t)96 * w.n1pad);
    w.sim = c.take<float>((size_t)N0 * w.n1pad);
    w.rlse = c.take<float>(N0); w.clse = c.take<float>(w.n1pad); w.best0 = c.take<float>(N0); w.zeros = c.take<float>(w.n1pad);
    w.m0 = c.take<int32_t>(N0); w.m1 = c.take<int32_t>(w.n1pad);
    // attention partials: self (N0,N0)+(N1,N1) or cross (N0,N1)+(N1,N0) are in flight together
    const size_t pf = std::max(lg_attention_partial_floats(N0, N0) + lg_attention_partial_floats(N1, N1),
                               lg_attention_partial_floats(N0, N1) + lg_attention_partial_floats(N1, N0));
    w.part = c.take<float>(std::max<size_t>(pf, 1));
    w.ascratch = c.take<char>(lg_assign_scratch_bytes(N1));
    return (c.off + 255) / 256 * 256;
}
__global__ void lg_init_kernel(int32_t* ind0, int n0, int32_t* count0, int32_t* ind1, int n1, int32_t* count1, float* zeros, int nz,
                               const int32_t* live0, const int32_t* live1) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g < n0) ind0[g] = g;
    if (g < n1) ind1[g] = g;
    if (g < nz) zeros[g] = 0.f;
    if (g == 0) {
        *count0 = live0 ? max(0, min(n0, *live0)) : n0;
        *count1 = live1 ? max(0, min(n1, *live1)) : n1;
    }
}
}  // namespace

size_t xfh_lg_workspace_bytes(int N0, int N1) {
    if (N0 <= 0 || N1 <= 0) return 0;
    LgWs w;
    return carve_lg(nullptr, N0, N1, w);
}

// one pair; live0/live1 (optional, device): number of valid rows (<= N0 / N1) when the lists are zero-padded to a capacity
static int lg_match_impl(xfh_lg_handle h, const float* kpts0, const float* desc0, int N0, const int32_t* live0, float W0, float H0,
                         const float* kpts1, const float* desc1, int N1, const int32_t* live1, float W1, float H1, float min_conf,
                         int prune_min_kpts, int64_t* matches, float* scores, int32_t* n_matches, void* workspace, hipStream_t st) {
    LgWs w;
    carve_lg(workspace, N0, N1, w);
    const int N[2] = {N0, N1};
    int bad = 0;
    // current (x, cos, sin, ind, n) of each set
    float *x[2], *cs[2], *sn[2]; int32_t *ind[2], *nn[2];
    for (int s = 0; s < 2; ++s) { LgSet& S = w.s[s]; x[s] = S.xa; cs[s] = S.csa; sn[s] = S.sna; ind[s] = S.inda; nn[s] = S.na; }
    lg_init_kernel<<<ceil_div(std::max(std::max(N0, N1), w.n1pad), 256), 256, 0, st>>>(ind[0], N0, nn[0], ind[1], N1, nn[1], w.zeros, w.n1pad, live0, live1);
    launch_lg_encode(kpts0, N0, W0, H0, h->wr, cs[0], sn[0], st);
    launch_lg_encode(kpts1, N1, W1, H1, h->wr, cs[1], sn[1], st);
    // both images per launch: y[s] = epi(x_in[s] . W^T + b)
    auto lin = [&](const LgLin& L, int epi, const float* const xin[2], int ldx, float* const yout[2], int ldy, const float* gamma, const float* beta) {
        LgLinSide sd[2];
        for (int s = 0; s < 2; ++s) sd[s] = LgLinSide{xin[s], ldx, yout[s], ldy, nn[s], N[s], cs[s], sn[s]};
        bad |= launch_lg_linear(L.w, L.b, L.k, L.n, epi, sd, 2, gamma, beta, st);
    };
    auto both = [](float* a, float* b, int off) { struct P { float* p[2]; } r{{a + off, b + off}}; return r; };
    {
        const float* din[2] = {desc0, desc1};
        lin(h->input_proj, LG_EPI_STORE, din, LG_IN, both(x[0], x[1], 0).p, 192, nullptr, nullptr);
    }
    auto ffn = [&](const LgFfn& F) {           // x[:, 0:96] += W3 gelu(LN(W0 [x | msg]))
        const float* xin[2] = {x[0], x[1]};
        lin(F.l0, LG_EPI_LNGELU, xin, 192, both(w.s[0].hid, w.s[1].hid, 0).p, 192, F.gamma, F.beta);
        const float* hin[2] = {w.s[0].hid, w.s[1].hid};
        lin(F.l3, LG_EPI_RESIDUAL, hin, 192, both(x[0], x[1], 0).p, 192, nullptr, nullptr);
    };
    auto attend = [&](bool cross, float scale) {
        LgAttSide sd[2];
        for (int s = 0; s < 2; ++s) {
            const int t = cross ? s ^ 1 : s;
            const float* kv = w.s[t].qkv;
            sd[s] = LgAttSide{w.s[s].qkv, cross ? kv : kv + 96, cross ? kv + 96 : kv + 192, w.s[s].att, nullptr, nn[s], nn[t], N[s], N[t], 1};
        }
        if (h->prof_on) {
            if (h->prof_used + 2 > h->prof_ev.size()) {
                hipEvent_t a, b;
                (void)hipEventCreate(&a); (void)hipEventCreate(&b);
                h->prof_ev.push_back(a); h->prof_ev.push_back(b);
            }
            (void)hipEventRecord(h->prof_ev[h->prof_used], st);
        }
        launch_lg_attention(sd, 2, 288, 288, 288, 96, w.part, scale, st);
        if (h->prof_on) {
            (void)hipEventRecord(h->prof_ev[h->prof_used + 1], st);
            h->prof_used += 2;
            for (int s = 0; s < 2; ++s) h->prof_flops += 4.0 * N[s] * N[cross ? s ^ 1 : s] * LG_D;      // Q.K^T and P.V at the capacities
        }
    };
    const float self_scale = 1.0f / std::sqrt((float)LG_D);
    for (int i = 0; i < LG_LAYERS; ++i) {
        const LgLayer& L = h->layer[i];
        const float* xin[2] = {x[0], x[1]};
        const float* ain[2] = {w.s[0].att, w.s[1].att};
        // self block: [q|k|v] projection with the rotary embedding in its epilogue
        lin(L.wqkv, LG_EPI_ROTARY, xin, 192, both(w.s[0].qkv, w.s[1].qkv, 0).p, 288, nullptr, nullptr);
        attend(false, self_scale);
        lin(L.out_proj, LG_EPI_STORE, ain, 96, both(x[0], x[1], 96).p, 192, nullptr, nullptr);
        ffn(L.self_ffn);
        // cross block: [qk | v] projection (qk pre-scaled by 96^-1/4 on both sides)
        lin(L.to_qkv, LG_EPI_STORE, xin, 192, both(w.s[0].qkv, w.s[1].qkv, 0).p, 288, nullptr, nullptr);
        attend(true, 1.0f);
        lin(L.to_out, LG_EPI_STORE, ain, 96, both(x[0], x[1], 96).p, 192, nullptr, nullptr);
        ffn(L.cross_ffn);
        if (i == LG_LAYERS - 1 || prune_min_kpts >= (1 << 30)) continue;
        // width pruning: matchability > 1 - width_confidence (0.95), for a set that still holds more than prune_min_kpts
        LgRowSide rs[2];
        LgPruneSide ps[2];
        for (int s = 0; s < 2; ++s) {
            LgSet& S = w.s[s];
            const bool a = x[s] == S.xa;
            rs[s] = LgRowSide{x[s], S.z, nn[s], N[s]};
            ps[s] = LgPruneSide{S.z, nn[s], N[s], S.map, a ? S.nb : S.na, x[s], a ? S.xb : S.xa, cs[s], a ? S.csb : S.csa,
                                sn[s], a ? S.snb : S.sna, ind[s], a ? S.indb : S.inda};
        }
        launch_lg_dot(rs, 2, 192, L.match_w, L.match_b, st);
        launch_lg_prune(ps, 2, 0.05f, prune_min_kpts, 192, st);
        for (int s = 0; s < 2; ++s) { x[s] = ps[s].xo; cs[s] = ps[s].cso; sn[s] = ps[s].sno; ind[s] = ps[s].indo; nn[s] = ps[s].n_out; }
    }
    // assignment of the last layer
    const LgLayer& L = h->layer[LG_LAYERS - 1];
    {
        const float* xin[2] = {x[0], x[1]};
        lin(h->final_proj, LG_EPI_STORE, xin, 192, both(w.s[0].md, w.s[1].md, 0).p, 96, nullptr, nullptr);
        LgRowSide rs[2] = {LgRowSide{x[0], w.s[0].z, nn[0], N0}, LgRowSide{x[1], w.s[1].z, nn[1], N1}};
        launch_lg_dot(rs, 2, 192, L.match_w, L.match_b, st);
    }
    launch_lg_transpose(w.s[1].md, 96, nn[1], N1, w.md1t, w.n1pad, st);
    {
        LinSrc src{};
        src.x = w.s[0].md; src.ldx = 96;
        bad |= launch_linear_mfma(w.md1t, w.zeros, LG_D, w.n1pad, w.n1pad, false, LOAD_ROWMAJOR, src, N0, nn[0], w.sim, w.n1pad, st);
    }
    launch_lg_assign(w.sim, w.n1pad, nn[0], N0, nn[1], N1, w.s[0].z, w.s[1].z, w.rlse, w.clse, w.m0, w.m1, w.best0, ind[0], ind[1], min_conf,
                     matches, scores, n_matches, w.ascratch, st);
    if (bad) return xfh_set_error(XFH_ERR_UNSUPPORTED, "xfh_lg_match: missing linear kernel instantiation");
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return xfh_set_error(XFH_ERR_HIP, "xfh_lg_match: %s", hipGetErrorString(e));
    return XFH_OK;
}

int xfh_lg_match(xfh_lg_handle h, const float* kpts0, const float* desc0, int N0, float W0, float H0, const float* kpts1, const float* desc1,
                 int N1, float W1, float H1, float min_conf, int prune_min_kpts, int64_t* matches, float* scores, int32_t* n_matches,
                 void* workspace, size_t workspace_bytes, xfh_stream stream) {
    if (!h || !kpts0 || !desc0 || !kpts1 || !desc1 || !matches || !scores || !n_matches) return xfh_set_error(XFH_ERR_ARG, "xfh_lg_match: NULL argument");
    if (N0 <= 0 || N1 <= 0 || N0 > 16384 || N1 > 16384) return xfh_set_error(XFH_ERR_ARG, "xfh_lg_match: key-point counts must be in 1..16384");
    if (((size_t)desc0 | (size_t)desc1) & 15) return xfh_set_error(XFH_ERR_ARG, "xfh_lg_match: descriptors must be 16-byte aligned");
    const size_t need = xfh_lg_workspace_bytes(N0, N1);
    if (!workspace || workspace_bytes < need || ((size_t)workspace & 255)) return xfh_set_error(XFH_ERR_WORKSPACE, "xfh_lg_match: workspace too small or misaligned (%zu < %zu)", workspace_bytes, need);
    return lg_match_impl(h, kpts0, desc0, N0, nullptr, W0, H0, kpts1, desc1, N1, nullptr, W1, H1, min_conf, prune_min_kpts, matches, scores,
                         n_matches, workspace, (hipStream_t)stream);
}

int xfh_lg_match_pairs(xfh_lg_handle h, const float* kpts, const float* desc, const int32_t* counts, int P, int cap, float W, float H,
                       float min_conf, int prune_min_kpts, int64_t* matches, float* scores, int32_t* n_matches, void* workspace,
                       size_t workspace_bytes, xfh_stream stream) {
    if (!h || !kpts || !desc || !counts || !matches || !scores || !n_matches) return xfh_set_error(XFH_ERR_ARG, "xfh_lg_match_pairs: NULL argument");
    if (P <= 0 || cap <= 0 || cap > 16384) return xfh_set_error(XFH_ERR_ARG, "xfh_lg_match_pairs: need P > 0 and 1 <= cap <= 16384");
    if ((size_t)desc & 15) return xfh_set_error(XFH_ERR_ARG, "xfh_lg_match_pairs: descriptors must be 16-byte aligned");
    const size_t need = xfh_lg_workspace_bytes(cap, cap);
    if (!workspace || workspace_bytes < need || ((size_t)workspace & 255)) return xfh_set_error(XFH_ERR_WORKSPACE, "xfh_lg_match_pairs: workspace too small or misaligned (%zu < %zu)", workspace_bytes, need);
    for (int p = 0; p < P; ++p) {           // pairs run back to back on the stream and share the workspace
        const size_t f0 = 2 * (size_t)p, f1 = f0 + 1;
        const int rc = lg_match_impl(h, kpts + f0 * cap * 2, desc + f0 * cap * 64, cap, counts + f0, W, H, kpts + f1 * cap * 2, desc + f1 * cap * 64,
                                     cap, counts + f1, W, H, min_conf, prune_min_kpts, matches + (size_t)p * cap * 2, scores + (size_t)p * cap,
                                     n_matches + p, workspace, (hipStream_t)stream);
        if (rc != XFH_OK) return rc;
    }
    return XFH_OK;
}
