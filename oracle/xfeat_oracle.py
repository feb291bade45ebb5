"""CPU oracle for the XFeat inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``accelerated_features_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
use it, and only as the checker / the reported CPU baseline.

What it is: a functional restatement (plain functions over a ``state_dict`` of CPU tensors,
no ``nn.Module``) of the arithmetic the reference executes on its CPU path, written from
the reference's behaviour:

    XFeatModel.forward          /root/reference/modules/model.py:123-154
    BasicLayer                  /root/reference/modules/model.py:12-25
    InterpolateSparse2d         /root/reference/modules/interpolator.py:10-33
    XFeat.detectAndCompute      /root/reference/modules/xfeat.py:49-103
    XFeat.get_kpts_heatmap      /root/reference/modules/xfeat.py:242-247
    XFeat.NMS                   /root/reference/modules/xfeat.py:249-263
    XFeat.match / batch_match   /root/reference/modules/xfeat.py:327-348 / 265-290
    XFeat.extractDense/dualscale/root/reference/modules/xfeat.py:356-394
    XFeat.refine_matches        /root/reference/modules/xfeat.py:306-325
    XFeat.subpix_softmax2d      /root/reference/modules/xfeat.py:292-304
    XFeat.preprocess_tensor     /root/reference/modules/xfeat.py:219-240

Pinning: the reference ships no tests or golden vectors and its trained
weights are absent, so the oracle is pinned against the reference ITSELF: ``tests/golden/
make_golden.py`` imports ``/root/reference/modules`` unmodified (CPU), runs it on the seeded
synthetic fixture of ``tests/fixtures.py`` and commits the outputs under ``tests/golden/``;
``tests/test_oracle_golden.py`` checks this file against those vectors.

The sparse sampling routines are written out as explicit fp32 arithmetic (they are the
contract the HIP kernels replicate); they are verified against
``torch.nn.functional.grid_sample`` in ``tests/test_oracle_sampling.py``.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


# --------------------------------------------------------------------------------------
# backbone
# --------------------------------------------------------------------------------------
def _basic(sd, name, x, stride=1, k=3):
    """conv(no bias) -> eval BatchNorm(no affine) -> ReLU  (model.py:16-22)."""
    y = F.conv2d(x, sd[f"{name}.layer.0.weight"], None, stride, k // 2)
    y = F.batch_norm(y, sd[f"{name}.layer.1.running_mean"], sd[f"{name}.layer.1.running_var"],
                     None, None, False, 0.0, BN_EPS)
    return torch.relu(y)


def _plain(sd, name, x):
    return F.conv2d(x, sd[f"{name}.weight"], sd[f"{name}.bias"])


def unfold8(x):
    """(B,1,H,W) -> (B,64,H/8,W/8), channel = 8*dy+dx  (model.py:113-120 with ws=8)."""
    B, C, H, W = x.shape
    assert C == 1
    t = x.reshape(B, H // 8, 8, W // 8, 8)          # b, i, dy, j, dx
    return t.permute(0, 2, 4, 1, 3).reshape(B, 64, H // 8, W // 8)


def backbone(sd, x, keep=False):
    """x: (B,C,H,W) float32, H and W multiples of 32.

    Returns (feats (B,64,H/8,W/8), logits (B,65,H/8,W/8), reliability (B,1,H/8,W/8)) and,
    with keep=True, a dict of every intermediate activation as 4th element.
    """
    t = {}
    g = x.mean(dim=1, keepdim=True)                                   # model.py:135
    g = F.instance_norm(g, eps=1e-5)                                  # model.py:35,136
    t["gray"] = g
    a = _basic(sd, "block1.0", g, 1); t["block1.0"] = a
    a = _basic(sd, "block1.1", a, 2); t["block1.1"] = a
    a = _basic(sd, "block1.2", a, 1); t["block1.2"] = a
    a = _basic(sd, "block1.3", a, 2); t["block1.3"] = a
    s = _plain(sd, "skip1.1", F.avg_pool2d(g, 4, 4)); t["skip1"] = s  # model.py:40-41
    a = a + s; t["x1"] = a                                            # model.py:140
    a = _basic(sd, "block2.0", a); t["block2.0"] = a
    a = _basic(sd, "block2.1", a); t["block2.1"] = a
    x3 = _basic(sd, "block3.0", a, 2); t["block3.0"] = x3
    x3 = _basic(sd, "block3.1", x3); t["block3.1"] = x3
    x3 = _basic(sd, "block3.2", x3, 1, 1); t["block3.2"] = x3
    x4 = _basic(sd, "block4.0", x3, 2); t["block4.0"] = x4
    x4 = _basic(sd, "block4.1", x4); t["block4.1"] = x4
    x4 = _basic(sd, "block4.2", x4); t["block4.2"] = x4
    x5 = _basic(sd, "block5.0", x4, 2); t["block5.0"] = x5
    x5 = _basic(sd, "block5.1", x5); t["block5.1"] = x5
    x5 = _basic(sd, "block5.2", x5); t["block5.2"] = x5
    x5 = _basic(sd, "block5.3", x5, 1, 1); t["block5.3"] = x5
    hw = x3.shape[-2:]
    u4 = F.interpolate(x4, tuple(hw), mode="bilinear")                # model.py:146
    u5 = F.interpolate(x5, tuple(hw), mode="bilinear")                # model.py:147
    f = x3 + u4 + u5; t["pyramid"] = f
    f = _basic(sd, "block_fusion.0", f); t["block_fusion.0"] = f
    f = _basic(sd, "block_fusion.1", f); t["block_fusion.1"] = f
    feats = _plain(sd, "block_fusion.2", f); t["feats"] = feats
    h = _basic(sd, "heatmap_head.0", feats, 1, 1); t["heatmap_head.0"] = h
    h = _basic(sd, "heatmap_head.1", h, 1, 1); t["heatmap_head.1"] = h
    rel = torch.sigmoid(_plain(sd, "heatmap_head.2", h)); t["reliability"] = rel
    kx = unfold8(g); t["unfold"] = kx                                 # model.py:152
    kx = _basic(sd, "keypoint_head.0", kx, 1, 1); t["keypoint_head.0"] = kx
    kx = _basic(sd, "keypoint_head.1", kx, 1, 1); t["keypoint_head.1"] = kx
    kx = _basic(sd, "keypoint_head.2", kx, 1, 1); t["keypoint_head.2"] = kx
    logits = _plain(sd, "keypoint_head.3", kx); t["logits"] = logits
    if keep:
        return feats, logits, rel, t
    return feats, logits, rel


# --------------------------------------------------------------------------------------
# pre-processing
# --------------------------------------------------------------------------------------
def preprocess(x):
    """float32 (B,C,H,W) -> resized to multiples of 32, plus (rh, rw)  (xfeat.py:233-240)."""
    x = x.float()
    H, W = x.shape[-2:]
    _H, _W = (H // 32) * 32, (W // 32) * 32
    rh, rw = H / _H, W / _W
    x = F.interpolate(x, (_H, _W), mode="bilinear", align_corners=False)
    return x, rh, rw


def parse_input(x):
    """xfeat.py:396-403: add batch dim; numpy HWC -> NCHW / 255."""
    if len(x.shape) == 3:
        x = x[None, ...]
    if isinstance(x, np.ndarray):
        x = torch.tensor(x).permute(0, 3, 1, 2) / 255
    return x


# --------------------------------------------------------------------------------------
# keypoint heat map + NMS
# --------------------------------------------------------------------------------------
def kpts_heatmap(logits):
    """softmax over the 65 channels, drop the dustbin, depth-to-space 8x8 (xfeat.py:242-247):
    heat[b,0,8i+dy,8j+dx] = softmax(logits[b,:,i,j])[8*dy+dx]."""
    p = F.softmax(logits, 1)[:, :64]
    B, _, h, w = p.shape
    p = p.reshape(B, 8, 8, h, w)                      # b, dy, dx, i, j
    return p.permute(0, 3, 1, 4, 2).reshape(B, 1, h * 8, w * 8)


def nms(heat, threshold=0.05, kernel_size=5):
    """xfeat.py:249-263.  Returns a list (len B) of int64 (n_b, 2) arrays of (x, y),
    row-major order (y outer, x inner)."""
    pad = kernel_size // 2
    local_max = F.max_pool2d(heat, kernel_size, 1, pad)
    pos = (heat == local_max) & (heat > threshold)
    out = []
    for b in range(heat.shape[0]):
        yx = pos[b, 0].nonzero()
        out.append(torch.stack([yx[:, 1], yx[:, 0]], -1))
    return out


def pad_keypoints(lists):
    """zero-pad the ragged NMS lists to (B, Nmax, 2) int64 (xfeat.py:256-261)."""
    n = max(len(k) for k in lists)
    out = torch.zeros((len(lists), n, 2), dtype=torch.long)
    for b, k in enumerate(lists):
        out[b, : len(k)] = k
    return out


# --------------------------------------------------------------------------------------
# sparse sampling: explicit fp32 arithmetic of InterpolateSparse2d + grid_sample
#   (interpolator.py:17-19,31-32; ATen GridSampler, align_corners=False, zeros padding)
# --------------------------------------------------------------------------------------
def sample_coords(pos, H, W, Hm, Wm):
    """pos (.., 2) integer or float (x, y) in an HxW frame -> (ux, uy) float32 in pixels of
    an Hm x Wm map:   g = 2*(p/(S-1)) - 1 (fp32) ;  u = fma(g+1, Sm/2, -0.5).

    The last step is ONE rounding: ATen's CPU grid sampler evaluates (g+1)*(Sm/2)-0.5 with a
    fused multiply-add (measured: the unfused form is off by up to 1e-5 in the sampled
    value, the fused form agrees to 4e-7).  Emulated here through float64."""
    p = pos.to(torch.float32)
    den = torch.tensor([W - 1, H - 1], dtype=torch.float32)
    g1 = (2.0 * (p / den) - 1.0) + 1.0                       # fp32, two roundings like the reference
    ux = (g1[..., 0].double() * (Wm * 0.5) - 0.5).float()
    uy = (g1[..., 1].double() * (Hm * 0.5) - 0.5).float()
    return ux, uy


def _gather2d(m, iy, ix):
    """m (C,Hm,Wm); iy, ix int64 (N,) -> (N,C) with zeros outside the map."""
    C, Hm, Wm = m.shape
    ok = (ix >= 0) & (ix < Wm) & (iy >= 0) & (iy < Hm)
    flat = m.reshape(C, -1)
    idx = (iy.clamp(0, Hm - 1) * Wm + ix.clamp(0, Wm - 1))
    v = flat[:, idx].t()
    return v * ok[:, None].to(m.dtype)


def sample_nearest(m, pos, H, W):
    """m (C,Hm,Wm), pos (N,2) -> (N,C).  nearest = round-half-to-even (nearbyint)."""
    ux, uy = sample_coords(pos, H, W, m.shape[1], m.shape[2])
    return _gather2d(m, torch.round(uy).long(), torch.round(ux).long())


def sample_bilinear(m, pos, H, W):
    ux, uy = sample_coords(pos, H, W, m.shape[1], m.shape[2])
    x0, y0 = torch.floor(ux), torch.floor(uy)
    tx, ty = ux - x0, uy - y0
    x0, y0 = x0.long(), y0.long()
    w00 = ((1 - tx) * (1 - ty))[:, None]
    w01 = (tx * (1 - ty))[:, None]
    w10 = ((1 - tx) * ty)[:, None]
    w11 = (tx * ty)[:, None]
    return (_gather2d(m, y0, x0) * w00 + _gather2d(m, y0, x0 + 1) * w01
            + _gather2d(m, y0 + 1, x0) * w10 + _gather2d(m, y0 + 1, x0 + 1) * w11)


def cubic_weights(t, A=-0.75):
    """Keys cubic convolution taps for offsets -1, 0, +1, +2 (ATen get_cubic_upsample_coefficients)."""
    def near(x):   # |x| <= 1
        return ((A + 2) * x - (A + 3)) * x * x + 1
    def far(x):    # 1 < |x| < 2
        return ((A * x - 5 * A) * x + 8 * A) * x - 4 * A
    return far(t + 1), near(t), near(1 - t), far(2 - t)


def sample_bicubic(m, pos, H, W):
    ux, uy = sample_coords(pos, H, W, m.shape[1], m.shape[2])
    x0, y0 = torch.floor(ux), torch.floor(uy)
    wx = cubic_weights(ux - x0)
    wy = cubic_weights(uy - y0)
    x0, y0 = x0.long(), y0.long()
    out = 0
    for j in range(4):
        row = 0
        for i in range(4):
            row = row + _gather2d(m, y0 - 1 + j, x0 - 1 + i) * wx[i][:, None]
        out = out + row * wy[j][:, None]
    return out


# --------------------------------------------------------------------------------------
# detectAndCompute (sparse)
# --------------------------------------------------------------------------------------
def detect_and_compute(sd, x, top_k=4096, detection_threshold=0.05, keep=False):
    """xfeat.py:49-103.  x float32 (B,C,H,W).  Returns list of dicts (keypoints (n,2) f32 in
    original-image pixels, scores (n,), descriptors (n,64)); with keep=True also a dict of
    stage outputs."""
    x, rh, rw = preprocess(x)
    B, _, H, W = x.shape
    feats, logits, rel = backbone(sd, x)
    fn = F.normalize(feats, dim=1)                                     # xfeat.py:70
    heat = kpts_heatmap(logits)
    cand = nms(heat, detection_threshold, 5)
    mk = pad_keypoints(cand)                                           # (B,N,2) int64
    N = mk.shape[1]
    scores = torch.empty((B, N), dtype=torch.float32)
    for b in range(B):
        sn = sample_nearest(heat[b], mk[b], H, W)[:, 0]
        sb = sample_bilinear(rel[b], mk[b], H, W)[:, 0]
        scores[b] = sn * sb                                            # xfeat.py:79
    scores[torch.all(mk == 0, dim=-1)] = -1                            # xfeat.py:80
    order = torch.argsort(-scores)                                     # xfeat.py:83
    mk = torch.gather(mk, 1, order[..., None].expand(-1, -1, 2))[:, :top_k]
    scores = torch.gather(scores, 1, order)[:, :top_k]
    desc = torch.stack([sample_bicubic(fn[b], mk[b], H, W) for b in range(B)])
    desc = F.normalize(desc, dim=-1)                                   # xfeat.py:93
    kp = mk * torch.tensor([rw, rh]).view(1, 1, -1)                    # xfeat.py:96
    valid = scores > 0
    out = [{"keypoints": kp[b][valid[b]], "scores": scores[b][valid[b]],
            "descriptors": desc[b][valid[b]]} for b in range(B)]
    if keep:
        return out, {"feats": feats, "logits": logits, "reliability": rel, "heat": heat,
                     "candidates": cand, "rh": rh, "rw": rw}
    return out


# --------------------------------------------------------------------------------------
# mutual nearest neighbour matching
# --------------------------------------------------------------------------------------
def match_mnn(d1, d2, min_cossim=0.82):
    """xfeat.py:327-348.  Returns (idx0, idx1) int64; ties resolve to the first maximum."""
    s12 = d1 @ d2.t()
    s21 = d2 @ d1.t()
    best12 = s12.max(dim=1)[1]
    best21 = s21.max(dim=1)[1]
    i0 = torch.arange(len(best12))
    keep = best21[best12] == i0
    if min_cossim > 0:
        keep = keep & (s12.max(dim=1)[0] > min_cossim)
    return i0[keep], best12[keep]


def batch_match(f1, f2, min_cossim=-1):
    """xfeat.py:265-290: one bmm, argmax both ways, mutual test per batch item."""
    s = torch.bmm(f1, f2.permute(0, 2, 1))
    m12 = torch.argmax(s, dim=-1)
    m21 = torch.argmax(s.permute(0, 2, 1), dim=-1)
    i0 = torch.arange(m12.shape[1])
    out = []
    for b in range(len(f1)):
        keep = m21[b][m12[b]] == i0
        if min_cossim > 0:
            keep = keep & (s[b].max(dim=1)[0] > min_cossim)
        out.append((i0[keep], m12[b][keep]))
    return out


# --------------------------------------------------------------------------------------
# semi-dense extraction + refinement
# --------------------------------------------------------------------------------------
def extract_dense(sd, x, top_k):
    """xfeat.py:356-377: top-k reliability cells, RAW (un-normalised) features, cell corner
    coordinates 8*(j,i) scaled by (rw, rh)."""
    if top_k < 1:
        top_k = 100_000_000
    x, rh, rw = preprocess(x)
    feats, _, rel = backbone(sd, x)
    B, C, h, w = feats.shape
    ii, jj = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    xy = (torch.stack([jj, ii], -1).reshape(-1, 2) * 8).expand(B, -1, -1)
    fm = feats.permute(0, 2, 3, 1).reshape(B, -1, C)
    r = rel.permute(0, 2, 3, 1).reshape(B, -1)
    _, idx = torch.topk(r, k=min(r.shape[1], top_k), dim=-1)
    f = torch.gather(fm, 1, idx[..., None].expand(-1, -1, 64))
    kp = torch.gather(xy, 1, idx[..., None].expand(-1, -1, 2))
    kp = kp * torch.tensor([rw, rh]).view(1, -1)
    return kp, f, idx


def extract_dualscale(sd, x, top_k, s1=0.6, s2=1.3):
    """xfeat.py:379-394."""
    x1 = F.interpolate(x, scale_factor=s1, align_corners=False, mode="bilinear")
    x2 = F.interpolate(x, scale_factor=s2, align_corners=False, mode="bilinear")
    k1, f1, _ = extract_dense(sd, x1, int(top_k * 0.20))
    k2, f2, _ = extract_dense(sd, x2, int(top_k * 0.80))
    kp = torch.cat([k1 / s1, k2 / s2], dim=1)
    sc = torch.cat([torch.ones(k1.shape[:2]) * (1 / s1), torch.ones(k2.shape[:2]) * (1 / s2)], dim=1)
    return kp, sc, torch.cat([f1, f2], dim=1)


def detect_and_compute_dense(sd, x, top_k=4096, multiscale=True):
    """xfeat.py:105-128."""
    x = x.float()
    if multiscale:
        kp, sc, f = extract_dualscale(sd, x, top_k)
    else:
        kp, f, _ = extract_dense(sd, x, top_k)
        sc = torch.ones(kp.shape[:2])
    return {"keypoints": kp, "descriptors": f, "scales": sc}


def fine_matcher(sd, v):
    """model.py:97-111: 4 x (Linear+bias -> eval BatchNorm1d(no affine) -> ReLU) -> Linear."""
    for li, bi in ((0, 1), (3, 4), (6, 7), (9, 10)):
        v = F.linear(v, sd[f"fine_matcher.{li}.weight"], sd[f"fine_matcher.{li}.bias"])
        v = F.batch_norm(v, sd[f"fine_matcher.{bi}.running_mean"], sd[f"fine_matcher.{bi}.running_var"],
                         None, None, False, 0.0, BN_EPS)
        v = torch.relu(v)
    return F.linear(v, sd["fine_matcher.12.weight"], sd["fine_matcher.12.bias"])


def subpix_softmax2d(o, temp=3):
    """xfeat.py:292-304: expectation of (x-4, y-4) under softmax(temp*o) on the 8x8 grid,
    flattened index = 8*y + x.  o: (n,64) -> (n,2)."""
    p = torch.softmax(temp * o, -1)
    idx = torch.arange(64)
    gx = (idx % 8 - 4).to(p.dtype)
    gy = (idx // 8 - 4).to(p.dtype)
    return torch.stack([(p * gx).sum(1), (p * gy).sum(1)], -1)


def refine_matches(sd, d0, d1, matches, batch_idx, fine_conf=0.25):
    """xfeat.py:306-325 -> (n',4) float32 rows (x0,y0,x1,y1)."""
    idx0, idx1 = matches[batch_idx]
    f1 = d0["descriptors"][batch_idx][idx0]
    f2 = d1["descriptors"][batch_idx][idx1]
    k0 = d0["keypoints"][batch_idx][idx0].clone()
    k1 = d1["keypoints"][batch_idx][idx1]
    sc0 = d0["scales"][batch_idx][idx0]
    o = fine_matcher(sd, torch.cat([f1, f2], dim=-1))
    conf = F.softmax(o * 3, dim=-1).max(dim=-1)[0]
    k0 = k0 + subpix_softmax2d(o) * sc0[:, None]
    good = conf > fine_conf
    return torch.cat([k0[good], k1[good]], dim=-1)


def match_xfeat_star(sd, im1, im2, top_k=4096):
    """xfeat.py:188-217 (always returns the list form)."""
    o1 = detect_and_compute_dense(sd, parse_input(im1), top_k)
    o2 = detect_and_compute_dense(sd, parse_input(im2), top_k)
    idxs = batch_match(o1["descriptors"], o2["descriptors"])
    return [refine_matches(sd, o1, o2, idxs, b) for b in range(len(idxs))]


def match_xfeat(sd, im1, im2, top_k=4096, min_cossim=-1, detection_threshold=0.05):
    """xfeat.py:165-186 -> (kpts0 (n,2), kpts1 (n,2), idx0, idx1)."""
    o1 = detect_and_compute(sd, parse_input(im1).float(), top_k, detection_threshold)[0]
    o2 = detect_and_compute(sd, parse_input(im2).float(), top_k, detection_threshold)[0]
    i0, i1 = match_mnn(o1["descriptors"], o2["descriptors"], min_cossim)
    return o1["keypoints"][i0], o2["keypoints"][i1], i0, i1
