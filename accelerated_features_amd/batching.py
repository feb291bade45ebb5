"""Batched pair matching for evaluation runners.

The reference's benchmarks call `matcher_fn(img0, img1)` one pair at a time
(modules/eval/megadepth1500.py:199-237, scannet1500.py:255-300).  On an MI355X a single VGA pair
leaves the device almost idle; this runner takes the whole list of pairs, groups it by image size,
pushes each group through `XFeat._detect_device` + `XFeat.match_pairs_device` in batches of up to
`max_pairs`, and reads back one small tensor of counts per batch.  The results are what
`XFeat.match_xfeat` returns pair by pair, in the original order: an image's result does not depend on the batch it travels in
(every kernel choice of the library is by image size, never by batch size: tests/test_gpu_census.py).
"""
import numpy as np
import torch

from .sharding import shard_range


def _as_nchw(img):
    """One image -> (C,H,W) tensor and the divisor parse_input would apply (255 for uint8 numpy HWC)."""
    if isinstance(img, np.ndarray):
        if img.ndim == 2:
            img = img[..., None]
        if img.ndim != 3:
            raise RuntimeError('For numpy arrays, only (H,W) or (H,W,C) format is supported.')
        t = torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1)
        return (t, 255.0) if img.dtype == np.uint8 else (t / 255, None)
    if img.dim() == 4 and img.shape[0] == 1:
        img = img[0]
    if img.dim() != 3:
        raise RuntimeError('Input tensor needs to be in (C,H,W) or (1,C,H,W) format')
    return img, None


def match_pairs(xfeat, pairs, top_k=None, min_cossim=-1, max_pairs=32, rank=0, world=1):
    """pairs: sequence of (img0, img1), each a numpy (H,W[,C]) image (scaled by 1/255 like
    XFeat.parse_input) or a (C,H,W) / (1,C,H,W) tensor.  Returns a list (same order, only this rank's
    shard when world > 1) of (mkpts0, mkpts1) numpy float32 (N,2) arrays -- XFeat.match_xfeat's result.
    """
    from .xfeat import _U8Image
    if top_k is None:
        top_k = xfeat.top_k
    lo, hi = shard_range(len(pairs), rank, world)
    items = []
    for i in range(lo, hi):
        a, da = _as_nchw(pairs[i][0])
        b, db = _as_nchw(pairs[i][1])
        items.append((i, a, da, b, db))
    groups = {}
    for it in items:
        key = (tuple(it[1].shape), tuple(it[3].shape), it[1].dtype, it[3].dtype, it[2], it[4])
        groups.setdefault(key, []).append(it)
    out = {}
    for key, members in groups.items():
        same_shape = key[0] == key[1] and key[2] == key[3] and key[4] == key[5]
        for s in range(0, len(members), max_pairs):
            chunk = members[s:s + max_pairs]
            if same_shape:
                frames = torch.stack([t for it in chunk for t in (it[1], it[3])])        # (2P,C,H,W): frames 2i, 2i+1 = pair i
                x = _U8Image(frames, key[4]) if key[4] is not None else frames
                res = _detect_exact(xfeat, x, top_k)
                kpts, desc, nv = res
                idx0, idx1, nm = xfeat.match_pairs_device(desc, nv, min_cossim)
                _collect(out, chunk, kpts[0::2], kpts[1::2], idx0, idx1, nm)
            else:                                   # the two images of a pair differ in size: one batch per side
                xa = torch.stack([it[1] for it in chunk])
                xb = torch.stack([it[3] for it in chunk])
                xa = _U8Image(xa, key[4]) if key[4] is not None else xa
                xb = _U8Image(xb, key[5]) if key[5] is not None else xb
                ka, da, na = _detect_exact(xfeat, xa, top_k)
                kb, db, nb = _detect_exact(xfeat, xb, top_k)
                idx0, idx1, nm = xfeat.match_sets_device(da, na, db, nb, min_cossim)
                _collect(out, chunk, ka, kb, idx0, idx1, nm)
    return [out[i] for i in range(lo, hi)]


def _collect(out, chunk, kp0, kp1, idx0, idx1, nm):
    """Matched coordinates of a whole chunk in three device->host copies (not two per pair): gather on the device into
    fixed-capacity arrays, slice by the match counts on the host.  Entries past a pair's count index row 0 (harmless)."""
    n = nm.to(torch.int64)
    keep = torch.arange(idx0.shape[1], device=idx0.device)[None] < n[:, None]
    g0 = torch.gather(kp0, 1, torch.where(keep, idx0, 0)[..., None].expand(-1, -1, 2))
    g1 = torch.gather(kp1, 1, torch.where(keep, idx1, 0)[..., None].expand(-1, -1, 2))
    packed = torch.cat([g0, g1], -1).cpu().numpy()                 # (P, cap, 4)
    counts = nm.cpu().tolist()
    for p, it in enumerate(chunk):
        out[it[0]] = (packed[p, :counts[p], :2].copy(), packed[p, :counts[p], 2:].copy())


def _detect_exact(xfeat, x, top_k):
    """_detect_device with the capacity re-run of detectAndCompute (plateau images), results still on the device."""
    cap = None
    B = x.shape[0]
    while True:
        cnt = torch.zeros((3, B), dtype=torch.int32, device=xfeat.dev)      # n_valid, n_candidates, [2, 0] = the status word: one read-back for all of it
        with xfeat.net.status_into(cnt[2]):
            kpts, scores, desc, n_valid, n_cand, cap, hw = xfeat._detect_device(x, top_k, None, cap, counts_out=cnt[:2])
        host = cnt.cpu()
        ncmax = int(host[1].max())
        if xfeat.net.fx_range_exceeded(status=int(host[2, 0])):      # fp16-pair arithmetic out of range (never on images): exact re-run on the fp32-range kernels, like detectAndCompute
            continue
        if cap >= hw or ncmax <= cap:
            return kpts, desc, n_valid
        cap = min(hw, max(ncmax, 2 * cap))


def match_pairs_star(xfeat, pairs, top_k=None, max_pairs=16, rank=0, world=1):
    """The semi-dense matcher (`--matcher xfeat-star`, modules/eval/megadepth1500.py:265-269) over a list of pairs:
    pairs whose two images share one size are grouped by that size and pushed through XFeat.match_xfeat_star in batches
    of up to `max_pairs`; the rest go pair by pair.  Returns, in order, what match_xfeat_star(img0, img1) returns for one
    pair: (mkpts0, mkpts1) numpy float32 (N,2)."""
    lo, hi = shard_range(len(pairs), rank, world)
    items, groups, out = [], {}, {}
    for i in range(lo, hi):
        a, da = _as_nchw(pairs[i][0])
        b, db = _as_nchw(pairs[i][1])
        a = a / da if da is not None else a            # the dense path interpolates first: convert like parse_input
        b = b / db if db is not None else b
        items.append((i, a.float(), b.float()))
    for it in items:
        key = (tuple(it[1].shape), tuple(it[2].shape))
        groups.setdefault(key, []).append(it)
    for key, members in groups.items():
        step = max_pairs if key[0] == key[1] else 1
        for s in range(0, len(members), step):
            chunk = members[s:s + step]
            xa = torch.stack([it[1] for it in chunk])
            xb = torch.stack([it[2] for it in chunk])
            res = xfeat.match_xfeat_star(xa, xb, top_k=top_k)
            if len(chunk) == 1:
                out[chunk[0][0]] = res                 # B == 1: already the (mkpts0, mkpts1) numpy tuple
            else:
                for it, m in zip(chunk, res):
                    m = m.cpu().numpy()
                    out[it[0]] = (m[:, :2], m[:, 2:])
    return [out[i] for i in range(lo, hi)]
