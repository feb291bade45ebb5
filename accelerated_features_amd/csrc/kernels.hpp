// Internal declarations shared by the kernel translation units and api.hip.
#pragma once
#include "common.hpp"

namespace xfh {

constexpr int GS_CHUNKS = 64;   // partial-sum chunks per image in gray_stats_kernel

// Index of every conv in execution order == accelerated_features_amd/spec.py::CONVS.
enum Layer {
    L_SKIP1 = 0, L_BLOCK1_0, L_BLOCK1_1, L_BLOCK1_2, L_BLOCK1_3, L_BLOCK2_0, L_BLOCK2_1,
    L_BLOCK3_0, L_BLOCK3_1, L_BLOCK3_2, L_BLOCK4_0, L_BLOCK4_1, L_BLOCK4_2,
    L_BLOCK5_0, L_BLOCK5_1, L_BLOCK5_2, L_BLOCK5_3, L_FUSION_0, L_FUSION_1, L_FUSION_2,
    L_HEAT_0, L_HEAT_1, L_HEAT_2, L_KP_0, L_KP_1, L_KP_2, L_KP_3, L_NUM
};

struct ConvW {
    int cin, cout, ks, stride, relu, cout_pad;   // cout_pad = cout rounded up to 32
    const float* w_oihw;   // (cout,cin,k,k)  BN folded
    const float* w_kc;     // [(ci*kk+tap)][cout]      BN folded (block1 direct kernels)
    const float* w_kcp;    // [(ci*kk+tap)][cout_pad]  BN folded, zero padded (MFMA kernels)
    const float* bias;     // [cout_pad] folded BN shift or conv bias, zero padded
    const void* w_fx;      // the fp16-pair fragments (weight_split.hpp: split_weight) in the operand order of the layer's kernel -- the 24-channel layers (k_conv_bx.hip:
                           // [step][fragment][64 lanes][8]) and the stride-2 64-channel layers (pack_bx64) --, NULL if the layer has none or a weight is too large for the pair
    const void* w_rs;      // 64 -> 64 / 128 -> 128 3x3 stride-1 layers: the fp16-pair image in conv_rs64_kernel's order (pack_rs64 / pack_rs128); the 1x1 behind one: pack_rs64_1x1; else NULL
};

struct LinW {             // fine_matcher layer: y = relu?(x W^T + b), BN folded
    int k, n, n_pad, relu;
    const float* w_kn;    // [k][n_pad]
    const float* bias;    // [n_pad]
    const void* w_fx;     // the same weights as fp16-pair fragments in linear_fx_kernel's operand order (weight_split.hpp: pack_linear_fx), NULL if a weight is too large for the pair
};

struct NetWeights {
    ConvW conv[L_NUM];
    LinW fine[5];
    const float* zeros;   // 1 KiB of zeros (padding source of the LDS-DMA loaders)
    // heads on the fp16 matrix cores (k_heads.hip: head_bx_kernel): [0] key-point head, [1] reliability head
    const void* head_fx[2];          // per layer [K step 4][cout block][fragment 3][64 lanes][8] fp16, or NULL (a weight beyond the pair's range: the head runs on head_f32r_kernel)
    const float* head_bx_bias[2];    // biases padded to the cout blocks (KP 64,64,64,96 ; REL 64,64)
    float head_rel_b_last;           // bias of the final 64 -> 1 layer of the reliability head
    float head_kp_b_dust;            // bias of the dustbin logit (output 64 of keypoint_head.3)
    const void* block1_fx;           // block1.3 in the fp16-pair arithmetic, compact LDS image (block1_fx.hpp), or NULL
    const void* block1_fx3;          // block1.2 likewise (q0 / q2 fragments)
};

struct Profiler;   // api.hip
extern int g_debug_cold;      // debug (xfh_debug_cold_start): MFMA kernels invalidate the instruction cache when they start (api.hip)

// Per-handle kernel switches (xfh_set_option; include/xfeat_hip.h documents them).  No process-wide state: a handle carries its own copy.
struct Options {
    int match_exact = 0;    // 1: xfh_match_mnn runs the exact f32-MFMA kernel for every pair (no filter)
    int match_sweep = 0;    // the filter's sweep: 0 = by shape (one orientation per tile when its wave tasks fill the chip, else two), 1 = two orientations, 2 = one
    int fx = 1 | 2 | 8 | 2048;      // XFH_FX_ALL: which layer families run in the fp16-pair arithmetic (a cleared bit: the f32-MFMA kernel of the family)
    int resize2 = 1;        // the fused two-stage resize of the dual-scale dense path: 1 = the tile's input region staged in LDS by 16-byte loads, 0 = four-byte gathers
    int block1 = 7;         // 7: block1.2 and block1.3 on the fp16 matrix cores; 5: the vector-ALU kernel
};

// ---- k_preproc.hip ----------------------------------------------------------------------
// gray = channel mean (raw), coef[b] = {alpha, beta} of the instance norm x = fmaf(gray, alpha, beta)
void launch_gray_norm(const float* img, int B, int C, int H, int W, double* part, float* gray, float* coef, hipStream_t st);
int launch_gray_norm_resized(const float* img, int B, int C, int Hin, int Win, int Hm, int Wm, float s1h, float s1w, int Ho, int Wo,
                             float s2h, float s2w, double* part, float* gray, float* coef, hipStream_t st, int form = 1);
void launch_gray_norm_u8(const unsigned char* img, bool nhwc, float divisor, int B, int C, int H, int W, double* part, float* gray,
                         float* coef, hipStream_t st);
void launch_resize_bilinear(const float* src, int planes, int Hin, int Win, float* dst, int Hout, int Wout,
                            float sh, float sw, hipStream_t st);
int launch_pyramid53(const ConvW& c53, const float* x3, const float* x4, const float* y5, float* out, int B,
                     int H3, int W3, int H4, int W4, int H5, int W5, hipStream_t st);      // block5.3 + the pyramid sum in one launch; -1: not this kernel's case
void launch_pyramid_sum(const float* x3, const float* x4, const float* x5, float* out, int planes,
                        int H3, int W3, int H4, int W4, int H5, int W5, hipStream_t st);

// ---- k_conv_direct.hip ------------------------------------------------------------------
void launch_block1(const NetWeights& nw, const float* gray, int B, int H, int W, float* t0, float* t1, float* t2,
                   float* x1, hipStream_t st);
void launch_block1_fused(const NetWeights& nw, const float* gray, const float* coef, int B, int H, int W, float* x1, hipStream_t st, int variant = 7, int* status = nullptr);
int launch_block1_layer(const NetWeights& nw, int layer, const float* in, int B, int Hin, int Win, float* out,
                        hipStream_t st);
void launch_conv_generic(const ConvW& c, const float* in, int B, int Hin, int Win, float* out, hipStream_t st);

// ---- k_conv_mfma.hip --------------------------------------------------------------------
// 3x3 / 1x1 convolution as an implicit GEMM on f32 MFMA.  in NCHW; out NCHW or NHWC.
// Returns 0, or -1 when no instantiation exists for the layer shape.
// fused1x1 (optional): the 1x1 conv that follows, computed in the same kernel.  zeros: >= 256 B.
int launch_conv_mfma(const ConvW& c, const ConvW* fused1x1, const float* zeros, const float* in, int B, int Hin, int Win,
                     float* out, bool nhwc_out, hipStream_t st, long long* trace = nullptr);
// the 24-channel 3x3 layers (stride 1 and 2) in the fp16-pair arithmetic (k_conv_bx.hip); -1 if no instantiation or no fp16-pair weights
int launch_conv_bx(const ConvW& c, const float* in, int B, int H, int W, float* out, hipStream_t st, long long* trace = nullptr, int* status = nullptr, bool in_cl = false, bool out_cl = false);
bool conv_bx_links(const ConvW& c, bool tracing);      // the layer can take / hand over channels-last activations
// 3x3/s1, 64 -> 64, fp16 pair, weights resident in registers (k_conv_rs64.hip / conv_rs64_body.hpp); -1: not this layer, or the map is wider than its LDS rings allow
int launch_conv_rs64(const ConvW& c, const float* in, int B, int H, int W, float* out, hipStream_t st, int* status = nullptr, const ConvW* fused1x1 = nullptr, bool nhwc = false,
                     long long* trace = nullptr);
bool conv_rs128_fits(int W);      // the map's rings fit into a CU's LDS
int launch_conv_rs128(const ConvW& c, const float* in, int B, int H, int W, float* out, hipStream_t st, int* status = nullptr);      // the 128 -> 128 form (block5.1, block5.2)
// 3x3/s2, 64 -> 64 | 128 (block4.0, block5.0) in the fp16-pair arithmetic (k_conv_bx64s2x.hip / conv_bx64s2_body.hpp); -1 if the layer has no fp16-pair weights
int launch_conv_bx64s2_fx(const ConvW& c, const float* in, int B, int H, int W, float* out, hipStream_t st, long long* trace = nullptr, int* status = nullptr);
int bx_steps(int cin);      // K steps of 16 = 2 groups of 8 channels of one tap
// ---- k_homography.hip (RANSAC + MAGSAC++ homography from match lists) ----
size_t homography_workspace_bytes(int P, int max_iters);
void launch_homography_tables(double thr, unsigned* stab, double* wtab, hipStream_t st);
int launch_find_homography(const float* p0, const float* p1, const int64_t* idx0, const int64_t* idx1, int kcap, const int32_t* counts, int n_const, int P, int cap, double thr, int max_iters,
                           double confidence, unsigned long long seed, double* H, unsigned char* mask, int32_t* info, void* ws, hipStream_t st);
double conv_flops(const ConvW& c, int B, int Hout, int Wout);

// ---- k_linear_mfma.hip ------------------------------------------------------------------
enum LinLoader { LOAD_ROWMAJOR = 0, LOAD_UNFOLD8 = 1, LOAD_GATHER2 = 2 };
struct LinSrc {
    const float* x;        // ROWMAJOR: (M,K) ; UNFOLD8: normalised gray (B,H,W) ; GATHER2: desc0 (P,N,64)
    int ldx;               // ROWMAJOR: row stride in floats
    int H, W;              // UNFOLD8: image size (M = B*(H/8)*(W/8))
    const float* x2;       // GATHER2: desc1 (P,N,64)
    const int64_t* idx0;   // GATHER2: (P,N)
    const int64_t* idx1;
    const int32_t* rowmap; // GATHER2: compact row -> p*N + r
    int N;                 // GATHER2: capacity per pair
};
// y (M,n) row-major with leading dimension ldy.  m_dev (optional) = device int32 with the live
// row count (<= M): workgroups past it exit.
int launch_linear_mfma(const float* w_kn, const float* bias, int K, int N, int n_pad, bool relu,
                       LinLoader loader, const LinSrc& src, int M, const int32_t* m_dev,
                       float* y, int ldy, hipStream_t st);
// the same layer in the fp16-pair arithmetic (linear_fx_body.hpp): three fp16 MFMAs per K = 16 instead of eight f32 MFMAs, fp32-equivalent results.  in_pair / out_pair: the
// rows are in the split form [ld halves xh | ld halves xl] (the bytes of an fp32 row of ld values) -- the fine_matcher's chain: 128 (fp32: ROWMAJOR / GATHER2) -> pair ->
// ... -> pair -> fp32.  n_pad a multiple of 64.  status: range guard of the pair.  -1: no instantiation
int launch_linear_fx(const void* w_fx, const float* bias, int K, int N, int n_pad, bool relu, LinLoader loader, const LinSrc& src, int M, const int32_t* m_dev,
                     void* y, int ldy, hipStream_t st, int* status, bool in_pair, bool out_pair);
// reliability = sigmoid(x . w + b) per row of x (M,64)                     (model.py:82-83)
void launch_dot_sigmoid(const float* x, int M, const float* w, const float* b, float* out, hipStream_t st);
// heat (B,H,W) <- softmax over 65 logits per cell, depth-to-space 8x8       (xfeat.py:242-247)
void launch_softmax_heat(const float* logits, int B, int hc, int wc, float* heat, hipStream_t st);

// ---- k_heads.hip -------------------------------------------------------------------------
// fused heads (persistent, weights LDS-resident): key-point head -> heat (+ optional logits (M,65)),
// reliability head -> sigmoid map
void launch_kp_head(const NetWeights& nw, const float* gray, const float* coef, int B, int H, int W, float* heat, float* logits, hipStream_t st, bool f32_kernels = false, int* status = nullptr);
// invnorm (optional): 1 / max(||feats[cell,:]||, 1e-12) per cell, a by-product of the layer-1 operand loads
void launch_rel_head(const NetWeights& nw, const float* feats, int ncell, float* reliab, float* invnorm, hipStream_t st, bool f32_kernels = false, int* status = nullptr);

// ---- k_detect.hip -----------------------------------------------------------------------
struct DetectWs {          // carved from the caller's workspace by api.hip
    unsigned long long* mask;   // (B, H, WPR) NMS flags, one bit per pixel
    int* wcount;                // (B, H*WPR)  popcounts
    unsigned* cand;             // (B, cap)    (y<<16)|x in row-major order
    unsigned long long* keys;   // (B, cap)    sort keys
    unsigned long long* skeys;  // (B, top_k)  sorted keys (~ord(score) << 32 | candidate slot)
    int* nsel;                  // (B)         min(n_candidates, cap, top_k)
    float* invnorm;             // (B, hc*wc)  1/max(||feats||,1e-12)
};
void launch_detect(const DetectWs& ws, const float* heat, const float* reliab, const float* feats, const float* invnorm, int B, int H, int W,
                   float thr, int top_k, int cap, float rw, float rh, float* kpts, float* scores, float* desc,
                   int32_t* n_valid, int32_t* n_cand, hipStream_t st, uint16_t* desc16 = nullptr, Profiler* prof = nullptr);
void launch_nms_only(const DetectWs& ws, const float* heat, int B, int H, int W, float thr, int kernel_size, int cap, int64_t* xy,
                     int32_t* n_cand, hipStream_t st);
// k_sampler.hip: InterpolateSparse2d as a stand-alone op; mode 0 nearest, 1 bilinear, 2 bicubic
void launch_sample_sparse(const float* x, const float* pos, int B, int C, int Hm, int Wm, int N, int H, int W, int mode, float* out,
                          hipStream_t st);
// top-k of a (B,n) float array, descending (ties: lower index first).  keys scratch (B,n) u64,
// skeys (B,k) u64 receives the sorted keys (index = low 32 bits).
void launch_topk_desc(const float* vals, int B, int n, int k, unsigned long long* keys, unsigned long long* skeys, int* nsel,
                      hipStream_t st);
void launch_dense_gather(const float* feats, const unsigned long long* skeys, int B, int hc, int wc, int k, float rw, float rh,
                         float scale_div, float* kpts, float* desc, int32_t* cell_index, hipStream_t st);

// ---- k_match.hip ------------------------------------------------------------------------
struct MatchWs {
    // zero-initialised per call (one memset): [rowkey | colkey | rowmaxh | colmaxh | nmax]
    void* zeroed; size_t zeroed_bytes;
    unsigned long long* rowkey;    // (P,N1) packed (ord(sim)<<32 | ~col): row arg-max, folded by 64-bit atomic max
    unsigned long long* colkey;    // (P,N2) packed (ord(sim)<<32 | ~row): column arg-max
    unsigned *rowmaxh, *colmaxh;   // (P,N1), (P,N2) ord(row / column maximum of the fp16 product), 32-bit atomic max across the sweep's workgroups
    unsigned* nmax;                // (2,P)  bit patterns of max |d1_i|, max |d2_j|
    // filter-and-refine scratch (k_match_f16.hip)
    _Float16 *a16, *b16;           // (P,N1,64), (P,N2,64) scaled fp16 copies
    float *na, *nb;                // (P,N1), (P,N2) fp32 norms
    float *thr_row, *thr_col;      // (P,N1), (P,N2) row / column maximum of the fp16 product - 2 E
    float *R, *C;                  // (P,ceil(N2/32),N1) / (P,ceil(N1/32),N2) block maxima of the fp16 product
};
void launch_match(const MatchWs& ws, const float* d1, size_t ps1, const float* d2, size_t ps2, const int32_t* n1,
                  const int32_t* n2, int n_stride, int n_off2, int P, int N1, int N2, float min_cossim,
                  int64_t* idx0, int64_t* idx1, int32_t* n_matches, hipStream_t st, Profiler* prof, const uint16_t* d1_16 = nullptr, const uint16_t* d2_16 = nullptr,
                  bool exact_only = false, int sweep_form = 0);
int match_row_blocks(int N1);
int match_debug_occupancy();

// ---- k_refine.hip -----------------------------------------------------------------------
void launch_refine_rowmap(const int32_t* n_matches, int P, int N, int32_t* offs, int32_t* rowmap, int32_t* total,
                          hipStream_t st);
void launch_refine_finish(const float* o /*(T,64)*/, const int32_t* rowmap, const int32_t* offs, const int32_t* total,
                          const float* kp0, const float* kp1, const float* scale0, const int64_t* idx0,
                          const int64_t* idx1, int P, int N, float fine_conf, float* out, int32_t* n_out,
                          float* rows_tmp, unsigned char* keep_tmp, hipStream_t st);

// ---- profiling hooks (api.hip) ------------------------------------------------------------
void prof_begin(Profiler* p, int which, hipStream_t st);
void prof_end(Profiler* p, int which, hipStream_t st, double flops, double bytes);

// ---- LighterGlue (k_lighterglue.hip): every per-set launcher takes the two images of the pair as `sides` ----
enum { LG_EPI_STORE = 0, LG_EPI_RESIDUAL = 1, LG_EPI_ROTARY = 2, LG_EPI_LNGELU = 3 };
struct LgLinSide { const float* x; int ldx; float* y; int ldy; const int32_t* n; int cap; const float* cs; const float* sn; };
struct LgAttSide { const float* Q; const float* K; const float* V; float* O; float* part; const int32_t* nq; const int32_t* nk; int qcap, kcap, nsplit; };
struct LgRowSide { const float* x; float* z; const int32_t* n; int cap; };
struct LgPruneSide {
    const float* z; const int32_t* n_in; int cap; int32_t* map; int32_t* n_out;
    const float* x; float* xo; const float* cs; float* cso; const float* sn; float* sno; const int32_t* ind; int32_t* indo;
};
void launch_lg_encode(const float* kpts, int N, float W, float H, const float* wr, float* cs, float* sn, hipStream_t st);
// wp: weights in operand order [N/32][2][K/8][32][4] (api_lg.hip packs them), bias (N).  Returns -1 if (K, epi) has no instantiation.
int launch_lg_linear(const float* wp, const float* bias, int K, int N, int epi, const LgLinSide* sides, int nsides, const float* gamma,
                     const float* beta, hipStream_t st);
size_t lg_attention_partial_floats(int qcap, int kcap);
// scratch: sum over sides of lg_attention_partial_floats(qcap, kcap) floats
void launch_lg_attention(LgAttSide* sides, int nsides, int ldq, int ldk, int ldv, int ldo, float* scratch, float scale, hipStream_t st);
void launch_lg_dot(const LgRowSide* sides, int nsides, int ld, const float* w, const float* b, hipStream_t st);
void launch_lg_prune(const LgPruneSide* sides, int nsides, float thr, int min_kpts, int ldx, hipStream_t st);
void launch_lg_transpose(const float* x, int ld, const int32_t* n_dev, int cap, float* xt, int npad, hipStream_t st);
size_t lg_assign_scratch_bytes(int cap1);
void launch_lg_assign(const float* sim, int ld, const int32_t* n0_dev, int cap0, const int32_t* n1_dev, int cap1, float* z0, float* z1,
                      float* rlse, float* clse, int32_t* m0, int32_t* m1, float* best0, const int32_t* ind0, const int32_t* ind1, float thr,
                      int64_t* matches, float* scores, int32_t* n_out, void* scratch, hipStream_t st);

}  // namespace xfh
