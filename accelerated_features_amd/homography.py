"""Robust homography from the matches on MI355X: the step after the matcher in the reference's demo.

    self.H, inliers = cv2.findHomography(points1, points2, cv2.USAC_MAGSAC, self.ransac_thr, maxIters=700, confidence=0.995)
    inliers = inliers.flatten() > 0                                          # /root/reference/realtime_demo.py:225-226

``find_homography`` has that call's arguments and results (H (3,3) float64 with H[2,2] = 1 and an (N,1) uint8 mask, or
``(None, None)`` when no model with at least four inliers exists); ``find_homography_batch`` is the same estimator over
P match lists resident in HBM (the output of ``XFeat.match_pairs_device`` / ``batch_match``) without leaving the device.
The kernels behind ``xfh_find_homography`` (include/xfeat_hip.h, csrc/k_homography.hip) evaluate every hypothesis at once
and apply RANSAC's stopping rule to the score list afterwards; OpenCV is not a dependency -- the estimator is the published
MAGSAC++ (Barath et al., CVPR 2020), so H agrees with cv2's as an estimate of the same homography, not in its random stream.
There is no CPU path: without the HIP library and a gfx950 device these functions raise.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

USAC_MAGSAC = 38          # cv2.USAC_MAGSAC: accepted (and required) as ``method`` for signature compatibility
INFO_FIELDS = ("found", "best_it", "iters", "n_inliers", "lo_accepted", "n", "score_lo", "score_hi")


def _device():
    if not torch.cuda.is_available():
        raise _lib.XFeatHipError("find_homography needs an AMD MI355X (gfx950) GPU; no CPU fallback exists")
    return torch.device('cuda', torch.cuda.current_device())


def _outputs(P, cap, dev):
    """H, mask, info: the select kernel writes every element (rows of the mask beyond a pair's count included), so no fill kernels."""
    return (torch.empty((P, 3, 3), dtype=torch.float64, device=dev), torch.empty((P, cap), dtype=torch.uint8, device=dev),
            torch.empty((P, 8), dtype=torch.int32, device=dev))


def _workspace(lib, P, max_iters, dev):
    ws = torch.empty(lib.xfh_homography_workspace_bytes(P, int(max_iters)) + 256, dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    ws.record_stream(torch.cuda.current_stream(dev))
    return ws, off


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def find_homography_batch(pts0, pts1, counts=None, ransac_thr=4.0, max_iters=700, confidence=0.995, seed=0):
    """P robust homographies in one call.

    pts0, pts1 : (P, cap, 2) float32 CUDA tensors (pixel coordinates of the matched key-points, row i of pts0 matches row i of pts1)
    counts     : (P,) int32 CUDA tensor, pair p uses its first counts[p] rows; None = all cap rows
    Returns a dict of CUDA tensors: 'H' (P,3,3) float64, 'inliers' (P,cap) uint8, 'info' (P,8) int32 (INFO_FIELDS).  Asynchronous.
    """
    dev = pts0.device if pts0.is_cuda else _device()
    pts0 = pts0.to(dev).float().contiguous()
    pts1 = pts1.to(dev).float().contiguous()
    if pts0.dim() != 3 or pts0.shape[2] != 2 or pts1.shape != pts0.shape:
        raise RuntimeError('expected two (P, cap, 2) point tensors of the same shape')
    P, cap = pts0.shape[0], pts0.shape[1]
    H, mask, info = _outputs(P, cap, dev)
    if P == 0 or cap == 0:
        return {'H': H, 'inliers': mask, 'info': info}
    if counts is not None:
        counts = counts.to(dev).to(torch.int32).contiguous()
        if counts.shape != (P,):
            raise RuntimeError('counts must have one entry per pair')
    lib = _lib.load()
    ws, off = _workspace(lib, P, max_iters, dev)
    _lib.check(lib.xfh_find_homography(_ptr(pts0), _ptr(pts1), _ptr(counts), cap, P, cap, float(ransac_thr), int(max_iters), float(confidence),
                                       int(seed) & ((1 << 64) - 1), _ptr(H), _ptr(mask), _ptr(info), C.c_void_p(ws.data_ptr() + off),
                                       ws.numel() - off, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "xfh_find_homography")
    return {'H': H, 'inliers': mask, 'info': info}


def find_homography_matches(kpts0, kpts1, idx0, idx1, n_matches, ransac_thr=4.0, max_iters=700, confidence=0.995, seed=0):
    """The same estimator straight on the matcher's output (no gathered point lists): correspondence i of pair p is
    (kpts0[p, idx0[p, i]], kpts1[p, idx1[p, i]]) for i < n_matches[p] -- ``points1 = kpts1[idx0]; points2 = kpts2[idx1]`` of
    realtime_demo.py:209-211.  kpts (P,K,2) float32, idx (P,cap) int64, n_matches (P,) int32: the CUDA tensors that
    ``XFeat._detect_device`` and ``XFeat.match_sets_device`` / ``match_pairs_device`` return.  Same result dict as find_homography_batch."""
    dev = kpts0.device
    if not kpts0.is_cuda:
        raise _lib.XFeatHipError("find_homography_matches works on device-resident match lists")
    P, cap = idx0.shape
    if kpts0.shape != kpts1.shape or kpts0.shape[0] != P or kpts0.shape[2] != 2 or idx1.shape != idx0.shape or n_matches.shape != (P,):
        raise RuntimeError('expected kpts (P,K,2), idx (P,cap), n_matches (P,)')
    for t, dt in ((kpts0, torch.float32), (kpts1, torch.float32), (idx0, torch.int64), (idx1, torch.int64), (n_matches, torch.int32)):
        if t.dtype != dt or not t.is_contiguous():
            raise RuntimeError('find_homography_matches: contiguous float32 key-points, int64 indices, int32 counts expected')
    H, mask, info = _outputs(P, cap, dev)
    lib = _lib.load()
    ws, off = _workspace(lib, P, max_iters, dev)
    _lib.check(lib.xfh_find_homography_matches(_ptr(kpts0), _ptr(kpts1), kpts0.shape[1], _ptr(idx0), _ptr(idx1), _ptr(n_matches), P, cap,
                                               float(ransac_thr), int(max_iters), float(confidence), int(seed) & ((1 << 64) - 1), _ptr(H),
                                               _ptr(mask), _ptr(info), C.c_void_p(ws.data_ptr() + off), ws.numel() - off,
                                               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "xfh_find_homography_matches")
    return {'H': H, 'inliers': mask, 'info': info}


class ReferenceTracker:
    """The XFeat branch of the reference demo's per-frame work (realtime_demo.py:136-142 and 204-231) for B camera streams at once, with
    everything resident in HBM: reference features are extracted once and cached (``ref_precomp``), every new frame goes through
    detectAndCompute -> mutual-NN match against the cached descriptors (``min_cossim`` 0.82) -> MAGSAC++ homography
    (``ransac_thr``, maxIters 700, confidence 0.995); a homography with fewer than ``min_inliers`` inliers is dropped (``self.H = None``)."""

    def __init__(self, xfeat, top_k=4096, min_cossim=0.82, ransac_thr=4.0, min_inliers=50, max_iters=700, confidence=0.995, seed=0):
        self.xfeat, self.top_k, self.min_cossim = xfeat, top_k, min_cossim
        self.ransac_thr, self.min_inliers, self.max_iters, self.confidence, self.seed = ransac_thr, min_inliers, max_iters, confidence, seed
        self.ref = None

    def set_reference(self, frames):
        """frames (B,C,H,W) (or (B,H,W,C) uint8 like the demo's camera frames): cache their key-points and descriptors."""
        kp, sc, de, nv, nc, cap, hw = self.xfeat._detect_device(self.xfeat.parse_input(frames), self.top_k)
        self.ref = (kp, de, nv.clone())
        self.ref_overflow = (nc, cap)          # device counts: see `overflowed`

    def track(self, frames):
        """One step for the B current frames.  Returns CUDA tensors: 'H' (B,3,3) float64, 'valid' (B,) bool (inliers >= min_inliers),
        'inliers' (B,top_k) uint8 over the match list, 'idx0' / 'idx1' (B,top_k) int64 (rows of the reference / current key-points),
        'n_matches' (B,) int32, 'keypoints' (B,top_k,2) of the current frames, 'info' (B,8).  No read-back happens here."""
        if self.ref is None:
            raise RuntimeError('ReferenceTracker.track: call set_reference first')
        kp0, de0, nv0 = self.ref
        kp1, sc, de1, nv1, nc, cap, hw = self.xfeat._detect_device(self.xfeat.parse_input(frames), self.top_k)
        if kp1.shape != kp0.shape:
            raise RuntimeError('the current frames must have the batch size of the reference frames')
        idx0, idx1, n = self.xfeat.match_sets_device(de0, nv0, de1, nv1, self.min_cossim)
        r = find_homography_matches(kp0, kp1, idx0, idx1, n, self.ransac_thr, self.max_iters, self.confidence, self.seed)
        r.update(valid=(r['info'][:, 0] > 0) & (r['info'][:, 3] >= self.min_inliers), idx0=idx0, idx1=idx1, n_matches=n, keypoints=kp1,
                 n_candidates=nc, nms_capacity=cap, fx_status=self.xfeat.net._status_target)      # fx_status: see `range_exceeded`
        return r

    def range_exceeded(self, fx_status):
        """True if an activation left the range of the fp16-pair arithmetic since the last check (device int32 `fx_status` of track()'s result, non-zero; never seen on
        images): the step's results are not valid -- the model has been switched to the fp32-range kernels, call set_reference / track again.  Costs a read-back, like
        `overflowed`: check it where the caller reads the step's results back anyway."""
        v = int(fx_status[0].item())
        if v:
            fx_status[:1].zero_()
        return self.xfeat.net.fx_range_exceeded(status=v)

    @staticmethod
    def overflowed(n_candidates, nms_capacity):
        """True if an image of the batch had more NMS candidates than the fixed capacity (plateau images): its candidate list was cut in
        row-major order, unlike detectAndCompute, which re-runs with room.  One read-back; check it when exactness on such images matters
        (``track()`` returns both values, ``set_reference`` leaves them in ``ref_overflow``)."""
        return bool((n_candidates > nms_capacity).any().item())


def find_homography(srcPoints, dstPoints, method=USAC_MAGSAC, ransacReprojThreshold=3.0, mask=None, maxIters=2000, confidence=0.995, *,
                    seed=0, return_info=False):
    """``cv2.findHomography(srcPoints, dstPoints, cv2.USAC_MAGSAC, ransacReprojThreshold, maxIters=..., confidence=...)`` for one pair:
    cv2's parameter names, order and defaults (``mask`` is cv2's optional output argument: accepted and ignored); ``method`` must be
    ``cv2.USAC_MAGSAC`` (38), the one the reference uses (realtime_demo.py:225).

    srcPoints, dstPoints : (N,2) or (N,1,2) arrays / tensors (numpy, CPU or CUDA torch), any float type
    Returns (H, inliers): H (3,3) float64 numpy array, inliers (N,1) uint8 numpy array -- or (None, None) like cv2 when fewer than
    four correspondences are given or no model is found.  ``seed`` fixes the sample sequence (same arguments, same bits).
    """
    if method != USAC_MAGSAC:
        raise _lib.XFeatHipError(f"find_homography: only cv2.USAC_MAGSAC ({USAC_MAGSAC}) is implemented, got method={method}")
    dev = _device()
    a = torch.as_tensor(np.asarray(srcPoints) if not torch.is_tensor(srcPoints) else srcPoints).reshape(-1, 2)
    b = torch.as_tensor(np.asarray(dstPoints) if not torch.is_tensor(dstPoints) else dstPoints).reshape(-1, 2)
    if a.shape != b.shape:
        raise RuntimeError('srcPoints and dstPoints must hold the same number of points')
    n = a.shape[0]
    if n < 4:
        return (None, None, dict.fromkeys(INFO_FIELDS, 0)) if return_info else (None, None)
    r = find_homography_batch(a.to(dev).float()[None], b.to(dev).float()[None], None, ransacReprojThreshold, maxIters, confidence, seed)
    info = dict(zip(INFO_FIELDS, r['info'][0].cpu().tolist()))
    if not info['found']:
        return (None, None, info) if return_info else (None, None)
    out = (r['H'][0].cpu().numpy(), r['inliers'][0].cpu().numpy().reshape(-1, 1))
    return out + (info,) if return_info else out
