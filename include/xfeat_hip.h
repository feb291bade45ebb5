/*
 * xfeat_hip.h -- C ABI of libxfeat_hip.so: the XFeat inference hot path on MI355X (gfx950).
 *
 * The reference (verlab/accelerated_features) is pure Python on PyTorch and has NO native
 * boundary; its drop-in boundary is the Python class modules/xfeat.py::XFeat.  This header
 * is the native boundary that class is re-hosted on: one entry point per block of ATen work
 * the reference's methods execute.  Each declaration cites the reference code it replaces.
 * A host in any language binds these symbols (ctypes stub: accelerated_features_amd/_lib.py,
 * other hosts: INTEGRATION.md).
 *
 * Conventions
 *   - Plain C types only.  All tensor pointers are DEVICE pointers (HBM) unless the name
 *     starts with host_.  Layouts are dense, row-major in the index order written.
 *   - The library never allocates in the hot path: the caller provides a workspace of
 *     xfh_*_workspace_bytes() bytes (256-byte aligned).  Only xfh_create allocates (weights).
 *   - Every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream).  No hidden synchronisation.  Process-wide state is limited to: the
 *     thread-local error string and per-device "kernel attribute set" flags (idempotent).  The library reads
 *     no environment variables; kernel-variant switches are per handle (xfh_set_option), and so are the
 *     debugging hooks xfh_debug_trace / xfh_profile_select (not for concurrent use on one handle).
 *   - Return value: XFH_OK (0) or a negative XFH_ERR_* code; xfh_last_error() gives a
 *     message for the calling thread.  No C++ exception crosses the boundary.
 *   - A handle's weights are immutable after xfh_create and it may be shared by threads/streams,
 *     provided each concurrent call uses its own workspace (the profiling hooks above write
 *     into the handle: switch them off for concurrent use).
 *   - Ragged results use fixed capacity + device-side counts: the host reads the counts back
 *     once per batch (the reference synchronises B times per batch: xfeat.py:254-261,99-103).
 */
#ifndef XFEAT_HIP_H
#define XFEAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XFH_VERSION 200          /* major*10000 + minor*100 + patch */

enum {
    XFH_OK = 0,
    XFH_ERR_ARG = -1,            /* bad shape / null pointer / unsupported size          */
    XFH_ERR_WEIGHTS = -2,        /* weight table malformed                               */
    XFH_ERR_WORKSPACE = -3,      /* workspace too small or misaligned                    */
    XFH_ERR_UNSUPPORTED = -4,    /* valid request outside what this build implements     */
    XFH_ERR_HIP = -5,            /* a HIP runtime call failed (message has the hipError) */
    XFH_ERR_DEVICE = -6          /* not a gfx950 device / no device                      */
};

typedef struct xfh_context* xfh_handle;
typedef void* xfh_stream;        /* hipStream_t */

int xfh_version(void);
const char* xfh_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Weights.  Replaces XFeat.__init__ / net.load_state_dict (modules/xfeat.py:23-35) and the
 * parameter containers of XFeatModel (modules/model.py:33-111).
 *
 * host_arrays: HOST pointers to the fp32 arrays of a reference state_dict in the canonical
 * order below (n_arrays must equal xfh_num_weight_arrays()).  Eval-mode BatchNorm
 * (affine=False, eps 1e-5) is folded into the preceding conv / linear at create time.
 *
 *   for each conv in execution order (accelerated_features_amd/spec.py::CONVS):
 *       BasicLayer : <name>.layer.0.weight (Cout,Cin,k,k), .layer.1.running_mean, .layer.1.running_var
 *       plain conv : <name>.weight (Cout,Cin,k,k), <name>.bias
 *   then fine_matcher: for L in 0,3,6,9: L.weight (out,in), L.bias, (L+1).running_mean, (L+1).running_var
 *                      then 12.weight, 12.bias
 * ---------------------------------------------------------------------------------------- */
int xfh_num_weight_arrays(void);
/* number of floats the i-th array must hold (for host-side validation) */
size_t xfh_weight_array_floats(int i);
int xfh_create(const float* const* host_arrays, int n_arrays, int device, xfh_handle* out);
void xfh_destroy(xfh_handle h);

/* ------------------------------------------------------------------------------------------
 * Bilinear resize, align_corners=False, NCHW planes.  Replaces F.interpolate in
 * preprocess_tensor (modules/xfeat.py:239, size given => scale = in/out) and in
 * extract_dualscale (modules/xfeat.py:380-381, scale_factor given => scale = 1/scale_factor).
 * src (planes,Hin,Win) -> dst (planes,Hout,Wout); scale_h/scale_w are the source-per-
 * destination steps PyTorch would use.
 * ---------------------------------------------------------------------------------------- */
int xfh_resize_bilinear(const float* src, int planes, int Hin, int Win, float* dst, int Hout, int Wout,
                        float scale_h, float scale_w, xfh_stream stream);

/* ------------------------------------------------------------------------------------------
 * Kernel switches of ONE handle.  Every layer has ONE default kernel and ONE fallback with fp32's range; the options choose between them (and two
 * test switches).  Results of either choice satisfy the same parity contract; the defaults are the shipped path.  Not for concurrent use with calls
 * on the same handle.
 *   "fx"            bitmask of XFH_FX_*, DEFAULT XFH_FX_ALL.  A set bit runs that layer family in the fp16-pair arithmetic on the fp16 matrix cores
 *                           (x = xh + 2^-11 xl, three fp16 MFMAs per product, fp32 accumulation: fp32-equivalent results for |x| < 65504, |w| < 31);
 *                           a cleared bit runs it on the f32 matrix cores (v_mfma_f32_32x32x2_f32: fp32's range).  A layer with a weight beyond the pair's range
 *                           runs on the f32 kernel whatever the bit says.
 *                             XFH_FX_CONV64 (1)    the 64- and 128-channel 3x3 convolutions: conv_rs64_kernel (weights resident in registers; block3.1 + .2, block4.1,
 *                                                  block4.2, block5.1, block5.2, block_fusion.0, block_fusion.1 + .2) and conv_bx64s2x_kernel (stride 2: block4.0, block5.0)
 *                             XFH_FX_CONV24 (2)    the 24-channel 3x3 convolutions (block2.0, block2.1, block3.0): conv_bx_kernel, conv_bxs2_kernel
 *                             XFH_FX_HEADS (8)     keypoint_head and heatmap_head: head_bx_kernel   (cleared: head_f32r_kernel)
 *                             XFH_FX_FINE (2048)   the fine_matcher's five linear layers (xfh_refine_matches, xfh_fine_matcher) as one chain: linear_fx_kernel (first / last
 *                                                  layer) and linear_fxd_kernel (the 512 -> 512 layers, LDS-DMA), the activations between them as fp16 pairs in the workspace
 *                                                  (csrc/linear_fx_body.hpp)   (cleared: linear_mfma_kernel)
 *                           Every other bit is rejected.
 *   "block1"        5 | 7   DEFAULT 7: block1.2 (8 -> 8) and block1.3 (8 -> 24, stride 2) on the fp16 matrix cores in the fp16-pair arithmetic (block1_mx_kernel);
 *                           5: every layer of block1 on the vector ALUs (block1_fused_kernel: fp32's range).  Both recompute conv1 inside conv2 (no c1 tile in LDS).
 *   "match_exact"   0 | 1   1: xfh_match_mnn computes every similarity on the f32 matrix cores (no fp16 filter): the kernel the default is tested against
 *   "match_sweep"   0..2    DEFAULT 0: the filter's pass over the fp16 product multiplies every 32 x 32 tile ONCE (mnn_f16_sweep2_kernel: column-direction block maxima over
 *                           the rows that meet in a lane) when P, N1, N2 give it enough wave tasks to fill the chip, else in both orientations (mnn_f16_sweep_kernel: finer
 *                           work split); 1 / 2: always the two- / one-orientation form.  The same matches in every case: decisions are taken on exact fp32 dot products.
 *   "resize2"       0 | 1   DEFAULT 1.  The fused two-stage resize of the dual-scale dense path (xfh_backbone_resized): 1 = the tile's input region staged in LDS by
 *                           16-byte loads (needs Win % 4 == 0; otherwise, and with 0: four-byte gathers per tap).  Same bits either way.
 * THE RANGE FALLBACK: on XFH_STATUS_FX_RANGE (below) repeat the call with fx = 0 and block1 = 5 -- every kernel then has fp32's range and none converts to fp16,
 * so the status bit cannot be set again.
 * Every kernel choice the options leave open is made by the IMAGE's size, never by the batch size: an image's results do not depend on the batch it travels in.
 * xfh_set_option returns XFH_ERR_ARG for an unknown key or value; xfh_get_option writes the current value.
 * ---------------------------------------------------------------------------------------- */
enum { XFH_FX_CONV64 = 1, XFH_FX_CONV24 = 2, XFH_FX_HEADS = 8, XFH_FX_FINE = 2048, XFH_FX_ALL = 1 | 2 | 8 | 2048 };
int xfh_set_option(xfh_handle h, const char* key, int value);
int xfh_get_option(xfh_handle h, const char* key, int* value);
/* Status word of a handle's calls: a caller-owned DEVICE int32 (NULL = none) into which kernels OR status bits; the caller zeroes and reads it (e.g. with
 * the read-back of the key-point counts).  XFH_STATUS_FX_RANGE: a convolution in the fp16-pair arithmetic (option "fx") met an activation of magnitude
 * >= 65504, which the fp16 high part cannot hold -- the outputs of that call are not valid; repeat it with fx = 0 and block1 = 5 (the f32-MFMA / vector-ALU kernels have fp32's range).
 * Set by xfh_backbone*, xfh_conv_layer, xfh_debug_block1, xfh_refine_matches, xfh_fine_matcher.  The pointer is read at launch time; it may be changed between calls. */
enum { XFH_STATUS_FX_RANGE = 1 };
int xfh_set_status_buffer(xfh_handle h, int32_t* device_word);

/* ------------------------------------------------------------------------------------------
 * Backbone.  Replaces XFeatModel.forward (modules/model.py:123-154) plus
 * XFeat.get_kpts_heatmap (modules/xfeat.py:242-247).
 *
 *   img      (B,C,H,W) fp32, H%32==0, W%32==0, C>=1; finite values.  A NaN or Inf pixel is NOT propagated the way F.relu / torch.max propagate it in the
 *            reference: block1's ReLU is a v_med3_f32 (k_conv_direct.hip is compiled with -fno-honor-nans, as is the matcher's maximum search), so a non-finite
 *            input gives unspecified finite-or-not values in that image (other images of the batch are unaffected).  Validate on the host if the source can produce them.
 *   feats    (B,H/8,W/8,64)  "M1", channels-last (a (B,64,h,w) tensor in channels_last memory format)
 *   logits   (B,H/8,W/8,65)  "K1", channels-last; may be NULL (then not written)
 *   heat     (B,H,W)         softmax(K1)[:64] depth-to-space 8x8; may be NULL (then logits must not be)
 *   reliab   (B,H/8,W/8)     "H1" after the sigmoid
 *   invnorm  (B,H/8,W/8)     optional (may be NULL): 1 / max(||feats[b,i,j,:]||, 1e-12), the per-cell factor of F.normalize(M1, dim=1)
 *                            (modules/xfeat.py:70) -- a by-product of the reliability head; hand it to xfh_detect_sparse to save that pass
 * ---------------------------------------------------------------------------------------- */
size_t xfh_backbone_workspace_bytes(int B, int C, int H, int W);
int xfh_backbone(xfh_handle h, const float* img, int B, int C, int H, int W,
                 float* feats, float* logits, float* heat, float* reliab, float* invnorm,
                 void* workspace, size_t workspace_bytes, xfh_stream stream);

/* Same network from uint8 pixels: replaces the host-side conversion in front of it --
 * XFeat.parse_input's `torch.tensor(x).permute(0,3,1,2) / 255` for numpy images (modules/xfeat.py:396-403, divisor 255,
 * layout XFH_LAYOUT_NHWC) and preprocess_tensor's `x.float()` for uint8 tensors (modules/xfeat.py:232, divisor 1,
 * layout XFH_LAYOUT_NCHW).  Per channel v = float(u8) / divisor; results are bit-identical to converting first. */
#define XFH_LAYOUT_NCHW 0
#define XFH_LAYOUT_NHWC 1
int xfh_backbone_u8(xfh_handle h, const uint8_t* img, int layout, float divisor, int B, int C, int H, int W,
                    float* feats, float* logits, float* heat, float* reliab, float* invnorm,
                    void* workspace, size_t workspace_bytes, xfh_stream stream);

/* The dual-scale dense path (XFeat.extract_dualscale, modules/xfeat.py:379-394): F.interpolate(x, scale_factor=s) to
 * (Hmid,Wmid), then preprocess_tensor's resize to multiples of 32 (modules/xfeat.py:234-238) to (Hout,Wout), then the
 * network.  Identical to xfh_resize_bilinear(img -> mid, scale1) + xfh_resize_bilinear(mid -> out, scale2) +
 * xfh_backbone(out) -- the gray plane the network consumes is bit-identical -- without writing either resized image.
 * img: (B,C,Hin,Win) fp32; workspace: xfh_backbone_workspace_bytes(B, C, Hout, Wout).  scale2 must be < 2
 * (XFH_ERR_UNSUPPORTED otherwise: materialise the images with xfh_resize_bilinear). */
int xfh_backbone_resized(xfh_handle h, const float* img, int B, int C, int Hin, int Win, int Hmid, int Wmid,
                         float scale1_h, float scale1_w, int Hout, int Wout, float scale2_h, float scale2_w,
                         float* feats, float* logits, float* heat, float* reliab, float* invnorm, void* workspace,
                         size_t workspace_bytes, xfh_stream stream);

/* One conv layer of the network in isolation (parity tests against per-layer oracle
 * activations).  layer = index into spec.CONVS; in (B,Cin,Hin,Win) NCHW, out (B,Cout,Hout,Wout)
 * NCHW with the layer's own stride/padding, folded BN and ReLU where the reference has them.
 * variant: XFH_CONV_VARIANT_*.  DEFAULT = the kernel the backbone uses for this layer under the handle's options; GENERIC = a plain direct convolution on the
 * vector ALUs (any layer: the independent device-side check); FX = the layer's fp16-pair kernel and F32 = its f32-MFMA kernel, each without fallback
 * (XFH_ERR_UNSUPPORTED where the layer has none); FX_PAIR / FX_PAIR_NHWC = this 3x3 layer AND the 1x1 layer behind it in one launch of conv_rs64_kernel
 * (layers block3.1, block_fusion.1: out = the 1x1's output, NCHW / channels-last). */
enum { XFH_CONV_VARIANT_DEFAULT = 0, XFH_CONV_VARIANT_GENERIC = 1, XFH_CONV_VARIANT_FX = 2, XFH_CONV_VARIANT_F32 = 3, XFH_CONV_VARIANT_FX_PAIR = 4, XFH_CONV_VARIANT_FX_PAIR_NHWC = 5 };
int xfh_conv_layer(xfh_handle h, int layer, const float* in, int B, int Hin, int Win, float* out,
                   int variant, xfh_stream stream);

/* ------------------------------------------------------------------------------------------
 * Sparse detection.  Replaces XFeat.NMS (modules/xfeat.py:249-263), the score / top-k /
 * descriptor block of detectAndCompute (modules/xfeat.py:70-103) and the three
 * InterpolateSparse2d calls (modules/interpolator.py:21-33).
 *
 *   heat (B,H,W), reliab (B,H/8,W/8), feats (B,H/8,W/8,64) as produced by xfh_backbone
 *   invnorm          (B,H/8,W/8) from the same xfh_backbone call, or NULL (then computed here in one extra pass over feats)
 *   threshold        detection threshold (strict >)
 *   top_k            any positive value (fewer candidates than top_k => shorter lists, like the reference)
 *   nms_capacity     capacity of the candidate list per image.  If n_candidates[b] comes back
 *                    larger, candidates were dropped in row-major order: re-run with a
 *                    capacity >= max(n_candidates) (H*W is always enough).
 *   rw, rh           original/processed size ratios (key-points are returned multiplied by them)
 * outputs (fixed capacity, entries past n_valid[b] are zero-filled):
 *   kpts (B,top_k,2) fp32 (x,y) ; scores (B,top_k) descending ; desc (B,top_k,64) unit norm
 *   desc_f16 (B,top_k,64) optional (may be NULL): fp16 (round-to-nearest-even) of 256 * desc, the operand of xfh_match_mnn's filter sweep
 *   n_valid (B) int32       = number of returned points with score > 0 (they form a prefix)
 *   n_candidates (B) int32  = NMS candidates found (uncapped)
 * ---------------------------------------------------------------------------------------- */
size_t xfh_detect_workspace_bytes(int B, int H, int W, int top_k, int nms_capacity);
int xfh_detect_sparse(xfh_handle h, const float* heat, const float* reliab, const float* feats, const float* invnorm,
                      int B, int H, int W, float threshold, int top_k, int nms_capacity, float rw, float rh,
                      float* kpts, float* scores, float* desc, uint16_t* desc_f16, int32_t* n_valid, int32_t* n_candidates,
                      void* workspace, size_t workspace_bytes, xfh_stream stream);

/* ------------------------------------------------------------------------------------------
 * Semi-dense extraction.  Replaces XFeat.extractDense after the backbone
 * (modules/xfeat.py:362-375): top-k of the reliability map (descending), RAW 64-D features of
 * those cells and their corner coordinates (8*(j,i)*(rw,rh)) / scale_div (scale_div = s of
 * extract_dualscale, modules/xfeat.py:388; 1 otherwise).
 *   kpts (B,k,2), desc (B,k,64), cell_index (B,k) int32 (may be NULL); k <= h*w
 * ---------------------------------------------------------------------------------------- */
size_t xfh_dense_workspace_bytes(int B, int h, int w, int k);
int xfh_extract_dense(xfh_handle h, const float* reliab, const float* feats, int B, int hc, int wc, int k,
                      float rw, float rh, float scale_div, float* kpts, float* desc, int32_t* cell_index,
                      void* workspace, size_t workspace_bytes, xfh_stream stream);

/* ------------------------------------------------------------------------------------------
 * Mutual nearest neighbour matching.  Replaces XFeat.match (modules/xfeat.py:327-348) and
 * XFeat.batch_match (modules/xfeat.py:265-290) for P independent pairs.
 *
 *   pair p uses d1 + p*pair_stride1 (N1 x 64, row-major) and d2 + p*pair_stride2 (N2 x 64)
 *   n1/n2: DEVICE int32 arrays; pair p uses n1[p*n_stride] rows and n2[p*n_stride+n_offset2]
 *          rows (so the n_valid array of xfh_detect_sparse can be passed for consecutive
 *          frame pairs with n_stride=2, n_offset2=1).  NULL => all N1 / N2 rows.
 *   d1_f16/d2_f16: optional (both or neither): fp16 round-to-nearest-even copies of 256 * d1 / 256 * d2 with the same pair strides (in elements,
 *          multiples of 8), valid ONLY for L2-normalised rows (|row| <= 1.00001) -- what xfh_detect_sparse's desc_f16 holds.  They spare the call its
 *          two conversion passes; every decision is still taken on exact fp32 dot products of d1 / d2 (results identical with or without them).
 *   The fp16 product only selects which 32-wide blocks of a row / column can hold its arg-max (window derived in k_match_f16.hip); the arg-max
 *   itself, ties included, comes from fp32 dot products of d1 / d2 in a fixed summation order.
 *   min_cossim <= 0 disables the similarity test (reference: `if min_cossim > 0`).
 *   pair_stride1/2 are in floats.
 * outputs: idx0, idx1 (P,N1) int64 (idx0 ascending), n_matches (P) int32.
 * Arg-max ties resolve to the lowest index, like torch.max / torch.argmax.
 * ---------------------------------------------------------------------------------------- */
size_t xfh_match_workspace_bytes(int P, int N1, int N2);
int xfh_match_mnn(xfh_handle h /* may be NULL */, const float* d1, size_t pair_stride1, const float* d2, size_t pair_stride2,
                  const uint16_t* d1_f16, const uint16_t* d2_f16,
                  const int32_t* n1, const int32_t* n2, int n_stride, int n_offset2,
                  int P, int N1, int N2, float min_cossim,
                  int64_t* idx0, int64_t* idx1, int32_t* n_matches,
                  void* workspace, size_t workspace_bytes, xfh_stream stream);

/* ------------------------------------------------------------------------------------------
 * Match refinement.  Replaces XFeat.refine_matches (modules/xfeat.py:306-325),
 * XFeat.subpix_softmax2d (modules/xfeat.py:292-304) and XFeatModel.fine_matcher
 * (modules/model.py:97-111) for P pairs at once.
 *
 *   desc0/desc1 (P,N,64), kp0/kp1 (P,N,2), scale0 (P,N)  -- detectAndComputeDense outputs
 *   idx0/idx1 (P,N) int64 with n_matches (P) int32 (device) -- xfh_match_mnn outputs
 *   out (P,N,4) fp32 rows (x0,y0,x1,y1) of the matches with conf > fine_conf, order kept;
 *   n_out (P) int32.
 * ---------------------------------------------------------------------------------------- */
size_t xfh_refine_workspace_bytes(int P, int N);
int xfh_refine_matches(xfh_handle h, const float* desc0, const float* desc1, const float* kp0, const float* kp1,
                       const float* scale0, const int64_t* idx0, const int64_t* idx1, const int32_t* n_matches,
                       int P, int N, float fine_conf, float* out, int32_t* n_out,
                       void* workspace, size_t workspace_bytes, xfh_stream stream);

/* ------------------------------------------------------------------------------------------
 * Stand-alone helpers behind the reference's public helper methods.
 *   xfh_kpts_heatmap : XFeat.get_kpts_heatmap (modules/xfeat.py:242-247); logits (B,h,w,65)
 *                      channels-last -> heat (B,8h,8w).
 *   xfh_nms          : XFeat.NMS (modules/xfeat.py:249-263), any odd kernel_size (the reference's generic
 *                      max_pool2d(k, 1, k/2) window; 5 is what the hot path uses and has the tiled kernel).
 *                      xy (B,capacity,2) int64 (x,y) in row-major order, zero padded; n_candidates (B) int32
 *                      uncapped.  workspace: xfh_detect_workspace_bytes(B,H,W,1,capacity).
 *   xfh_sample_sparse: InterpolateSparse2d.forward (modules/interpolator.py:10-33; the XFeat.interpolator attribute,
 *                      modules/xfeat.py:37): x (B,C,Hm,Wm) NCHW, pos (B,N,2) fp32 (x,y) in an H x W frame ->
 *                      out (B,N,C); grid_sample semantics (align_corners=False, zeros padding) with the reference's
 *                      fp32 coordinate arithmetic.  The hot path fuses its three sampling sites and does not call this.
 *   xfh_fine_matcher : XFeatModel.fine_matcher (modules/model.py:97-111): x (n,128) -> out (n,64).
 *                      workspace: xfh_refine_workspace_bytes(1, n).
 * ---------------------------------------------------------------------------------------- */
int xfh_kpts_heatmap(const float* logits, int B, int hc, int wc, float* heat, xfh_stream stream);
int xfh_nms(xfh_handle h, const float* heat, int B, int H, int W, float threshold, int kernel_size, int capacity, int64_t* xy,
            int32_t* n_candidates, void* workspace, size_t workspace_bytes, xfh_stream stream);
enum { XFH_SAMPLE_NEAREST = 0, XFH_SAMPLE_BILINEAR = 1, XFH_SAMPLE_BICUBIC = 2 };
int xfh_sample_sparse(const float* x, const float* pos, int B, int C, int Hm, int Wm, int N, int H, int W, int mode, float* out,
                      xfh_stream stream);
int xfh_fine_matcher(xfh_handle h, const float* x, int n, float* out, void* workspace, size_t workspace_bytes,
                     xfh_stream stream);

/* ------------------------------------------------------------------------------------------
 * Robust homography from the matches -- the consumer of the demo:
 *     H, inliers = cv2.findHomography(points1, points2, cv2.USAC_MAGSAC, ransac_thr, maxIters=700, confidence=0.995)
 * (realtime_demo.py:225), for P pairs at once.  OpenCV (opencv-contrib-python-headless 4.10.0.84, requirements.txt:1) is not
 * part of the reference tree: the algorithm is the published one (RANSAC + MAGSAC++ quality and sigma-consensus++ weights,
 * Barath et al., CVPR 2020) as specified in DESIGN.md 3.7 / csrc/k_homography.hip -- same estimate, not OpenCV's random stream.
 *   pts0/pts1 (P,cap,2) fp32 pixel coordinates (device), pair p uses its first counts[p] rows (device int32; NULL: n_const
 *   for all).  All max_iters (<= 4096) hypotheses are scored on the device; the stopping rule (confidence) is then applied
 *   to the score list as the sequential loop would apply it, so the result is a function of the arguments and `seed` only.
 *   H (P,9) fp64 row-major with H[8] = 1 (zeros when nothing was found); mask (P,cap) uint8, 1 = forward transfer error
 *   < ransac_thr; info (P,8) int32: found, winning hypothesis, hypotheses the loop would have run, inliers, accepted
 *   refinement steps, n, quality (lo, hi word).  Fewer than 4 correspondences / inliers: found = 0 (cv2 returns None).
 *   xfh_find_homography_matches: the same on the matcher's output without materialising the point lists: correspondence i of
 *   pair p is (kpts0[p][idx0[p][i]], kpts1[p][idx1[p][i]]), kpts (P,kpt_cap,2) fp32, idx (P,cap) int64 and n_matches (P) int32 as
 *   xfh_match_mnn writes them (realtime_demo.py:209-211: points1 = kpts1[idx0], points2 = kpts2[idx1]).
 *   xfh_homography_tables: the 4096-entry quality (20-bit fixed point) / weight tables over r^2 in [0, (2 thr)^2).
 * ---------------------------------------------------------------------------------------- */
size_t xfh_homography_workspace_bytes(int P, int max_iters);
int xfh_find_homography(const float* pts0, const float* pts1, const int32_t* counts, int n_const, int P, int cap,
                        double ransac_thr, int max_iters, double confidence, uint64_t seed,
                        double* H, uint8_t* mask, int32_t* info, void* workspace, size_t workspace_bytes, xfh_stream stream);
int xfh_find_homography_matches(const float* kpts0, const float* kpts1, int kpt_cap, const int64_t* idx0, const int64_t* idx1,
                                const int32_t* n_matches, int P, int cap, double ransac_thr, int max_iters, double confidence, uint64_t seed,
                                double* H, uint8_t* mask, int32_t* info, void* workspace, size_t workspace_bytes, xfh_stream stream);
int xfh_homography_tables(double ransac_thr, uint32_t* score_table, double* weight_table, xfh_stream stream);

/* ------------------------------------------------------------------------------------------
 * LighterGlue: the attention matcher of XFeat.match_lighterglue (modules/xfeat.py:131-162), i.e.
 * kornia.feature.lightglue.LightGlue.forward under the configuration of modules/lighterglue.py:12-27 (input 64-D,
 * d = 96, one head, 6 layers, no early stop, width pruning 0.95, filter threshold = min_conf), for ONE pair
 * (the reference asserts B = 1).  Its own checkpoint (weights/xfeat-lighterglue.pt) -> its own handle.
 *
 *   host_arrays : fp32 arrays in kornia's module order after the reference loader's renaming
 *                 (modules/lighterglue.py:41-46): input_proj.{weight,bias}, posenc.Wr.weight, per layer
 *                 transformers.i.self_attn.{Wqkv,out_proj,ffn.0,ffn.1,ffn.3}.{weight,bias},
 *                 transformers.i.cross_attn.{to_qk,to_v,to_out,ffn.0,ffn.1,ffn.3}.{weight,bias}, then
 *                 log_assignment.i.{matchability,final_proj}.{weight,bias}, then token_confidence.i.token.0.{weight,bias}
 *                 (xfh_lg_num_weight_arrays() = 169; sizes from xfh_lg_weight_array_floats).
 *   xfh_lg_match: kpts (N,2) pixel coordinates, desc (N,64), image size (W,H) per image.  prune_min_kpts: after every
 *                 layer but the last, a set that still has MORE than this many key-points drops those whose
 *                 matchability is <= 1 - width_confidence (the published pruning_keypoint_thresholds: -1 on CPU =
 *                 always, 1024 on CUDA, 1536 on CUDA with flash attention); XFH_LG_NO_PRUNING disables it.  Output: matches (<= min(N0,N1), 2) int64 indices into the input
 *                 lists, ascending in column 0; scores; n_matches (1) int32 -- all device memory, asynchronous.
 * ---------------------------------------------------------------------------------------- */
#define XFH_LG_NO_PRUNING (1 << 30)
typedef struct xfh_lg_context* xfh_lg_handle;
int xfh_lg_num_weight_arrays(void);
size_t xfh_lg_weight_array_floats(int index);
int xfh_lg_create(const float* const* host_arrays, int n_arrays, int device, xfh_lg_handle* out);
void xfh_lg_destroy(xfh_lg_handle h);
size_t xfh_lg_workspace_bytes(int N0, int N1);
int xfh_lg_match(xfh_lg_handle h, const float* kpts0, const float* desc0, int N0, float W0, float H0,
                 const float* kpts1, const float* desc1, int N1, float W1, float H1, float min_conf, int prune_min_kpts,
                 int64_t* matches, float* scores, int32_t* n_matches, void* workspace, size_t workspace_bytes,
                 xfh_stream stream);

/* bench.py hook: HIP events on the launch stream around every attention launch of this handle (lg_attention_kernel, both
 * images per launch) while enabled; xfh_lg_profile_read returns their number, summed duration and algorithmic FLOPs
 * (4 N_q N_k 96 per image and launch, at the key-point capacities -- run with XFH_LG_NO_PRUNING for exact figures) and resets. */
int xfh_lg_profile(xfh_lg_handle h, int enable);
int xfh_lg_profile_read(xfh_lg_handle h, int* n_launches, double* total_ms, double* total_flops);

/* The batch form used with xfh_detect_sparse's fixed-capacity outputs: frames (2p, 2p+1) of kpts (2P,cap,2) /
 * desc (2P,cap,64) / counts (2P) int32 (all device memory) form pair p; every image has size (W,H).  No host
 * read-back of the counts; pairs run back to back on `stream` and share one workspace of
 * xfh_lg_workspace_bytes(cap, cap).  Outputs: matches (P,cap,2) int64, scores (P,cap), n_matches (P) int32. */
int xfh_lg_match_pairs(xfh_lg_handle h, const float* kpts, const float* desc, const int32_t* counts, int P, int cap,
                       float W, float H, float min_conf, int prune_min_kpts, int64_t* matches, float* scores,
                       int32_t* n_matches, void* workspace, size_t workspace_bytes, xfh_stream stream);

/* ------------------------------------------------------------------------------------------
 * Kernel timing hooks for bench.py (HIP events recorded around one kernel family on the
 * launch stream).  which: see XFH_PROF_* ; xfh_profile_read synchronises the recorded events
 * and returns the number of launches and their summed duration in milliseconds.
 * ---------------------------------------------------------------------------------------- */
enum { XFH_PROF_NONE = 0, XFH_PROF_CONV_MFMA = 1, XFH_PROF_MATCH = 2, XFH_PROF_BLOCK1 = 3, XFH_PROF_HEADS = 4,
       XFH_PROF_CONV_64_64_S1 = 5 /* launches of conv_mfma_kernel<64,64,3,1,..>: the 64->64 3x3 stride-1 layers */,
       XFH_PROF_CONV_24_24 = 6 /* the two 24->24 3x3 stride-1 layers (block2.0 / block2.1): one kernel instantiation, two launches per step */,
       XFH_PROF_CONV_LAYER0 = 100 /* + index into spec.CONVS: one MFMA conv layer only */,
       XFH_PROF_ALL = 1000 /* every kernel (or tight kernel group) of xfh_backbone / xfh_detect_sparse / xfh_match_mnn as its own span: read with xfh_profile_read_spans */ };
/* span ids reported under XFH_PROF_ALL (next to XFH_PROF_BLOCK1 and XFH_PROF_CONV_LAYER0 + layer) */
enum { XFH_SPAN_GRAY = 200, XFH_SPAN_PYRAMID = 201, XFH_SPAN_HEAD_REL = 202, XFH_SPAN_HEAD_KP = 203,
       XFH_SPAN_NMS_FLAGS = 210, XFH_SPAN_NMS_COMPACT = 211, XFH_SPAN_TOPK = 212, XFH_SPAN_DESCRIPTOR = 213,
       XFH_SPAN_MATCH_ZERO = 220, XFH_SPAN_MATCH_PREP = 221, XFH_SPAN_MATCH_SWEEP = 222, XFH_SPAN_MATCH_REFINE = 223, XFH_SPAN_MATCH_FINALIZE = 224,
       XFH_SPAN_MATCH_EXACT = 225 };
int xfh_profile_select(xfh_handle h, int which);
/* the spans recorded since the last read, in launch order: ids[i], ms[i] for i < min(*n_spans, capacity); synchronises; resets */
int xfh_profile_read_spans(xfh_handle h, int* ids, double* ms, int capacity, int* n_spans);
/* debug: 24 int64 s_memtime stamps per MFMA-conv workgroup are written to device_buffer (NULL = off) */
int xfh_debug_trace(xfh_handle h, long long* device_buffer);
/* debug / torture (process-wide, not for production): with enable != 0 every matrix-core kernel of the backbone invalidates the instruction cache when a
 * workgroup starts, so that its first tile runs on instruction-fetch misses -- the condition under which round 3's split-bf16 key-point head (deleted in round 6) was found to
 * deliver a wrong 16-cell block; the concurrency / cold-start soaks of tests/test_gpu_parity.py, tools/final_soak.py and the code-position scan
 * (tools/bench_src/scan_probe.cpp) use it.
 * WHY THE HOOK SHIPS IN THE PRODUCTION LIBRARY instead of a debug build: the hazard it provokes depends on where a kernel's instructions lie relative to the
 * instruction-cache lines (3 of 16 code positions failed for that head).  Evidence gathered on a debug twin -- the same source compiled with another flag, i.e.
 * other offsets -- would say nothing about the shipped bytes.  The in-suite soak and the scans therefore torture THE library that ships; the price is one scalar
 * compare at kernel entry (cold == 0: not taken) and this one process-wide int, which no product path writes.  The xfh_debug_* entry points that only probe
 * (block1, trace, match_occupancy) are thin launchers of shipped kernels: they add no state. */
int xfh_debug_cold_start(int enable);
/* debug: block1 + skip1 alone in the form option "block1" selects: gray (B,H,W) raw gray image, coef (B,2) the per-image {alpha, beta} of the instance
 * normalisation (x -> alpha x + beta), x1 (B,24,H/4,W/4); H % 4 == W % 4 == 0.  For the variant-against-variant tests of tests/test_gpu_parity.py. */
int xfh_debug_block1(xfh_handle h, const float* gray, const float* coef, int B, int H, int W, float* x1, xfh_stream stream);
/* debug: resident workgroups per CU the runtime reports for mnn_sim_kernel */
int xfh_debug_match_occupancy(void);
int xfh_profile_read(xfh_handle h, int* n_launches, double* total_ms, double* total_flops, double* total_bytes);

#ifdef __cplusplus
}
#endif
#endif /* XFEAT_HIP_H */
