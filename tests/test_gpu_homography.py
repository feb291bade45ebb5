"""GPU parity of the homography consumer (/root/reference/realtime_demo.py:223-229) against
oracle/homography_oracle.py, through the C-ABI (xfh_find_homography / xfh_homography_tables).

Bit-exact: the winning hypothesis, the number of iterations the sequential loop would have run, the integer model quality and
the inlier mask.  Tolerance 1e-8 (relative to |H|): the refined homography (fp64 reductions in a different order)."""
import ctypes as C

import numpy as np
import pytest
import torch

import fixtures
from oracle import homography_oracle as ho
from test_oracle_homography import synthetic_pair, transfer_error

pytestmark = pytest.mark.gpu


def _info(t):
    from accelerated_features_amd.homography import INFO_FIELDS
    d = dict(zip(INFO_FIELDS, t.cpu().tolist()))
    d["score"] = (d.pop("score_hi") << 32) | (d.pop("score_lo") & 0xffffffff)
    return d


def _check_against_oracle(p0, p1, H, mask, info, thr, iters, conf, seed, pair=0):
    Ho, mo, io = ho.find_homography(p0, p1, thr, max_iters=iters, confidence=conf, seed=seed, pair=pair, return_info=True)
    for k in ("found", "best_it", "iters", "n_inliers", "score", "lo_accepted"):
        assert info[k] == io[k], (k, info, io)
    n = len(p0)
    if io["found"]:
        assert np.array_equal(mask[:n], mo[:, 0])
        assert np.abs(H - Ho).max() <= 1e-8 * np.abs(Ho).max(), np.abs(H - Ho).max()
        assert H[2, 2] == 1.0
    else:
        assert not H.any()
    assert not mask[n:].any()
    return io


def test_tables_are_the_oracles():
    from accelerated_features_amd import _lib as L
    lib = L.load()
    for thr in (0.75, 4.0, 11.5):
        st = torch.zeros(ho.NBINS, dtype=torch.int32, device="cuda")
        wt = torch.zeros(ho.NBINS, dtype=torch.float64, device="cuda")
        L.check(lib.xfh_homography_tables(thr, C.c_void_p(st.data_ptr()), C.c_void_p(wt.data_ptr()), None), "tables")
        torch.cuda.synchronize()
        _, so, wo = ho.tables(thr)
        assert np.array_equal(st.cpu().numpy().astype(np.uint32), so)           # closed forms (erf / erfc / exp) == scipy's incomplete gammas after rounding to 20 bits
        assert np.abs(wt.cpu().numpy() - wo).max() < 1e-12


@pytest.mark.parametrize("n,outliers,noise,thr,iters,conf,seed", [
    (4, 0.0, 0.0, 4.0, 700, 0.995, 0),
    (11, 0.0, 0.3, 4.0, 700, 0.995, 1),
    (300, 0.3, 0.5, 4.0, 700, 0.995, 2),
    (1000, 0.8, 1.0, 4.0, 700, 0.995, 3),
    (1500, 0.6, 1.0, 2.5, 1000, 0.999, 4),
    (4096, 0.45, 0.7, 4.0, 700, 0.995, 5),
    (777, 0.5, 0.8, 1.0, 257, 0.9, 2 ** 63 + 12345),
    (16384, 0.5, 0.6, 3.0, 4096, 0.9999, 7),
])
def test_single_pair_matches_oracle(n, outliers, noise, thr, iters, conf, seed):
    from accelerated_features_amd.homography import find_homography
    p0, p1, Ht, _ = synthetic_pair(n, outliers, noise, seed=100 + n)
    H, mask, info = find_homography(p0, p1, ransacReprojThreshold=thr, maxIters=iters, confidence=conf, seed=seed, return_info=True)
    info["score"] = (info.pop("score_hi") << 32) | (info.pop("score_lo") & 0xffffffff)
    assert H is not None and mask.shape == (n, 1) and mask.dtype == np.uint8 and H.dtype == np.float64
    io = _check_against_oracle(p0, p1, H, mask[:, 0], info, thr, iters, conf, seed)
    assert io["found"] == 1
    if noise > 0 and outliers < 0.8:
        assert transfer_error(H, Ht) < 1.0
    # same arguments, same bits
    H2, mask2 = find_homography(p0, p1, ransacReprojThreshold=thr, maxIters=iters, confidence=conf, seed=seed)
    assert np.array_equal(H, H2) and np.array_equal(mask, mask2)


def test_ragged_batch_matches_oracle_pair_by_pair():
    from accelerated_features_amd.homography import find_homography_batch
    counts = [0, 3, 4, 10, 700, 2048, 513, 1]
    cap = 2048
    p0 = np.zeros((len(counts), cap, 2), np.float32)
    p1 = np.zeros_like(p0)
    for p, n in enumerate(counts):
        if n:
            a, b, _, _ = synthetic_pair(n, 0.4 if n > 10 else 0.0, 0.5, seed=40 + p)
            p0[p, :n], p1[p, :n] = a, b
        p0[p, n:] = 1e9                                   # rows beyond the count must not be read as data
    r = find_homography_batch(torch.from_numpy(p0).cuda(), torch.from_numpy(p1).cuda(), torch.tensor(counts, dtype=torch.int32).cuda(),
                              ransac_thr=4.0, max_iters=700, confidence=0.995, seed=9)
    torch.cuda.synchronize()
    H, mask = r["H"].cpu().numpy(), r["inliers"].cpu().numpy()
    for p, n in enumerate(counts):
        info = _info(r["info"][p])
        assert info["n"] == n
        io = _check_against_oracle(p0[p, :n], p1[p, :n], H[p], mask[p], info, 4.0, 700, 0.995, 9, pair=p)
        assert io["found"] == (1 if n >= 4 else 0)


def test_degenerate_and_error_behaviour():
    from accelerated_features_amd import _lib as L
    from accelerated_features_amd.homography import find_homography, find_homography_batch
    assert find_homography(np.zeros((3, 2)), np.zeros((3, 2))) == (None, None)          # cv2: needs at least 4 points
    line = np.stack([np.arange(50.0), 3 * np.arange(50.0)], axis=1)
    assert find_homography(line, line + 1) == (None, None)                                  # every sample is collinear
    g = np.random.default_rng(0)
    a, b = g.uniform(0, 500, (200, 2)), g.uniform(0, 500, (200, 2))
    H, mask, info = find_homography(a, b, ransacReprojThreshold=1.0, maxIters=700, return_info=True)
    Ho, mo = ho.find_homography(a, b, 1.0)
    assert (H is None) == (Ho is None) and (H is None or np.array_equal(mask, mo))
    with pytest.raises(L.XFeatHipError):
        find_homography(a, b, method=8)                                                      # cv2.RANSAC: not this path
    with pytest.raises(L.XFeatHipError):
        find_homography(a, b, maxIters=5000)
    with pytest.raises(L.XFeatHipError):
        find_homography(a, b, confidence=1.0)
    r = find_homography_batch(torch.zeros(0, 16, 2), torch.zeros(0, 16, 2))
    assert r["H"].shape == (0, 3, 3)


def test_demo_pattern_cached_reference_then_match_then_homography():
    """realtime_demo.py:204-229: reference features cached, every frame detectAndCompute -> match -> findHomography.
    The frame is the reference texture translated by (64, 32) px (a multiple of the backbone's stride: the synthetic weights' descriptors are
    not trained for anything else) plus noise: the estimate must be that translation.  min_cossim is -1 instead of the demo's 0.82 for the
    same reason -- the list keeps its ~35 % wrong matches for RANSAC to reject."""
    from accelerated_features_amd import XFeat
    from accelerated_features_amd.homography import find_homography
    xf = XFeat(weights=fixtures.synthetic_state_dict(0), top_k=4096, detection_threshold=0.05)
    a, b = fixtures.shifted_pair(1, 480, 640, seed=7, shift=(32, 64), noise=0.01)
    ref = xf.detectAndCompute(a, top_k=4096)[0]
    cur = xf.detectAndCompute(b, top_k=4096)[0]
    idx0, idx1 = xf.match(ref["descriptors"], cur["descriptors"], -1)
    assert len(idx0) > 50
    points1 = ref["keypoints"][idx0].cpu().numpy()
    points2 = cur["keypoints"][idx1].cpu().numpy()
    from accelerated_features_amd.homography import USAC_MAGSAC
    H, inliers = find_homography(points1, points2, USAC_MAGSAC, 4.0, maxIters=700, confidence=0.995)      # the demo's call, verbatim
    inliers = inliers.flatten() > 0
    assert inliers.sum() > 0.4 * len(idx0)
    assert np.abs(H - np.array([[1, 0, 64.0], [0, 1, 32.0], [0, 0, 1]])).max() < 0.5 and abs(H[0, 0] - 1) < 5e-3 and abs(H[1, 0]) < 5e-3
    Ho, mo = ho.find_homography(points1, points2, 4.0)
    assert np.array_equal(mo[:, 0] > 0, inliers) and np.abs(H - Ho).max() <= 1e-8 * np.abs(Ho).max()


def test_index_list_entry_equals_gathered_points_and_tracker_runs_the_demo_step():
    """xfh_find_homography_matches reads kpts[idx] itself: same bits as xfh_find_homography on the gathered lists.  ReferenceTracker = the demo's
    per-frame step (cached reference, detect, match, homography) for B streams without a read-back."""
    from accelerated_features_amd import XFeat
    from accelerated_features_amd.homography import ReferenceTracker, find_homography_batch, find_homography_matches
    xf = XFeat(weights=fixtures.synthetic_state_dict(0), top_k=2048, detection_threshold=0.05)
    a, b = fixtures.shifted_pair(2, 256, 320, seed=9, shift=(32, 64), noise=0.01)
    tr = ReferenceTracker(xf, top_k=2048, min_cossim=-1, min_inliers=50, seed=3)
    with pytest.raises(RuntimeError):
        tr.track(b)
    tr.set_reference(a)
    r = tr.track(b)
    torch.cuda.synchronize()
    n = r["n_matches"].cpu().tolist()
    assert min(n) > 100 and r["valid"].cpu().tolist() == [True, True]
    H = r["H"].cpu().numpy()
    for p in range(2):
        assert np.abs(H[p] - np.array([[1, 0, 64.0], [0, 1, 32.0], [0, 0, 1]])).max() < 0.5
        m = r["inliers"][p].cpu().numpy()
        assert m[:n[p]].sum() == int(r["info"][p, 3]) >= 50 and not m[n[p]:].any()
    kp0 = tr.ref[0]
    p0 = torch.gather(kp0, 1, r["idx0"].clamp(0, 2047)[..., None].expand(-1, -1, 2)).contiguous()
    p1 = torch.gather(r["keypoints"], 1, r["idx1"].clamp(0, 2047)[..., None].expand(-1, -1, 2)).contiguous()
    g = find_homography_batch(p0, p1, r["n_matches"], 4.0, 700, 0.995, 3)
    for k in ("H", "inliers", "info"):
        assert torch.equal(g[k], r[k]), k
    again = find_homography_matches(kp0, r["keypoints"], r["idx0"], r["idx1"], r["n_matches"], 4.0, 700, 0.995, 3)
    assert torch.equal(again["H"], r["H"]) and torch.equal(again["inliers"], r["inliers"])
    # and against the oracle, pair 1 of the batch (the generator's counter carries the pair index)
    k = n[1]
    Ho, mo, io = ho.find_homography(p0[1, :k].cpu().numpy(), p1[1, :k].cpu().numpy(), 4.0, seed=3, pair=1, return_info=True)
    assert io["best_it"] == int(r["info"][1, 1]) and io["iters"] == int(r["info"][1, 2]) and np.array_equal(mo[:, 0], r["inliers"][1, :k].cpu().numpy())
    assert np.abs(H[1] - Ho).max() <= 1e-8 * np.abs(Ho).max()


def test_randomised_sweep_matches_oracle_and_repeats_bit_for_bit():
    """32 random configurations (list length, outlier share, noise, threshold, maxIters, confidence, seed, degenerate duplicates): every
    integer output identical to the restatement; then 100 repetitions of one call: identical bits (u64 atomics carry no order)."""
    from accelerated_features_amd.homography import find_homography, find_homography_batch
    g = np.random.default_rng(2024)
    for k in range(32):
        n = int(g.integers(4, 3000))
        outl, noise = float(g.uniform(0, 0.75)), float(g.uniform(0, 1.5))
        thr, iters = float(g.uniform(0.8, 6.0)), int(g.integers(20, 1500))
        conf, seed = float(g.choice([0.9, 0.99, 0.995, 0.9999])), int(g.integers(0, 2 ** 62))
        p0, p1, _, _ = synthetic_pair(n, outl, noise, seed=500 + k)
        if k % 5 == 0:                                   # repeated correspondences: duplicate draws, rank-deficient samples
            p0[: n // 2] = p0[0]
            p1[: n // 2] = p1[0]
        H, mask, info = find_homography(p0, p1, ransacReprojThreshold=thr, maxIters=iters, confidence=conf, seed=seed, return_info=True)
        info["score"] = (info.pop("score_hi") << 32) | (info.pop("score_lo") & 0xffffffff)
        if H is None:
            Ho, mo, io = ho.find_homography(p0, p1, thr, max_iters=iters, confidence=conf, seed=seed, return_info=True)
            assert Ho is None and info["best_it"] == io["best_it"] and info["iters"] == io["iters"], (k, info, io)
        else:
            _check_against_oracle(p0, p1, H, mask[:, 0], info, thr, iters, conf, seed)
    p0, p1, _, _ = synthetic_pair(2000, 0.5, 0.8, seed=77)
    a, b = torch.from_numpy(np.stack([p0] * 8)).cuda(), torch.from_numpy(np.stack([p1] * 8)).cuda()
    first = find_homography_batch(a, b, None, 3.0, 700, 0.995, 5)
    for _ in range(100):
        r = find_homography_batch(a, b, None, 3.0, 700, 0.995, 5)
        assert torch.equal(r["H"], first["H"]) and torch.equal(r["inliers"], first["inliers"]) and torch.equal(r["info"], first["info"])
    assert len({tuple(first["info"][p, 1:3].tolist()) for p in range(8)}) > 1      # pairs draw different samples (the counter carries the pair index)
