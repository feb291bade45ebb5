"""Tie-aware comparison helpers (finding 7).

Integer outputs of the path (key-point coordinates, match index pairs) are decided by
comparisons between fp32 numbers (NMS equality, detection threshold, top-k cut, arg-max).
Two correct fp32 evaluations of the network differ at the 1e-7 level (the reference's own
CPU path does between 1 and 8 threads), so "bit-exact" is defined as: identical integer
outputs, except for items whose deciding margin -- measured on the ORACLE's own float
maps -- is below a tiny epsilon.  Every exception is counted and must be explained by such a
margin; an unexplained difference fails the test.
"""
import numpy as np
import torch


def _as_np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def window_margin(heat, x, y, ks=5):
    """heat (H,W) np; margin of pixel (x,y) over the best OTHER pixel of its ks x ks window."""
    H, W = heat.shape
    r = ks // 2
    y0, y1, x0, x1 = max(0, y - r), min(H, y + r + 1), max(0, x - r), min(W, x + r + 1)
    win = heat[y0:y1, x0:x1].copy()
    c = win[y - y0, x - x0]
    win[y - y0, x - x0] = -np.inf
    return float(c - win.max())


def compare_keypoints(test, ref, heat=None, thr=0.05, rw=1.0, rh=1.0, score_tol=1e-4, desc_tol=1e-4,
                      eps_heat=2e-6, eps_score=2e-6, max_exceptions_frac=0.005):
    """test/ref: dicts with 'keypoints' (n,2), 'scores' (n,), 'descriptors' (n,64).

    Returns a report dict; raises AssertionError on an unexplained difference.
    heat: the oracle's (H,W) key-point heat map (needed to explain set differences).
    """
    kt, kr = _as_np(test["keypoints"]), _as_np(ref["keypoints"])
    st, sr = _as_np(test["scores"]), _as_np(ref["scores"])
    dt, dr = _as_np(test["descriptors"]), _as_np(ref["descriptors"])
    assert kt.ndim == 2 and kt.shape[1] == 2 and dt.shape[1:] == (64,)
    # scores must come out sorted (descending)
    assert np.all(np.diff(st) <= 0), "test scores are not sorted descending"
    key_t = {(float(x), float(y)): i for i, (x, y) in enumerate(kt)}
    key_r = {(float(x), float(y)): i for i, (x, y) in enumerate(kr)}
    assert len(key_t) == len(kt), "duplicate key-points in test output"
    common = [k for k in key_t if k in key_r]
    only_t = [k for k in key_t if k not in key_r]
    only_r = [k for k in key_r if k not in key_t]
    it = np.array([key_t[k] for k in common], dtype=np.int64)
    ir = np.array([key_r[k] for k in common], dtype=np.int64)
    rep = {"n_test": len(kt), "n_ref": len(kr), "common": len(common), "only_test": len(only_t),
           "only_ref": len(only_r)}
    if len(common):
        rep["score_maxdiff"] = float(np.abs(st[it] - sr[ir]).max())
        rep["desc_maxdiff"] = float(np.abs(dt[it] - dr[ir]).max())
        assert rep["score_maxdiff"] <= score_tol, rep
        assert rep["desc_maxdiff"] <= desc_tol, rep
        # rank agreement up to score ties
        rank_moved = it != ir
        if rank_moved.any():
            # a moved rank is fine when the scores in between are within tolerance
            lo, hi = np.minimum(it, ir), np.maximum(it, ir)
            span = np.abs(sr[np.minimum(hi, len(sr) - 1)] - sr[np.minimum(lo, len(sr) - 1)])
            rep["rank_moved"] = int(rank_moved.sum())
            rep["rank_moved_maxgap"] = float(span[rank_moved].max())
            assert rep["rank_moved_maxgap"] <= 10 * eps_score + 1e-6, rep
    # set differences must be explained by a near-tie in the oracle's own maps
    n_exc = len(only_t) + len(only_r)
    rep["exceptions"] = n_exc
    if n_exc:
        assert heat is not None, f"key-point sets differ and no heat map given: {rep}"
        heat = _as_np(heat)
        cut = float(min(sr.min(), st.min())) if len(sr) and len(st) else 0.0
        unexplained = []
        for (x, y) in only_t + only_r:
            xi, yi = int(round(x / rw)), int(round(y / rh))
            hv = float(heat[yi, xi])
            m = window_margin(heat, xi, yi)
            sc = st[key_t[(x, y)]] if (x, y) in key_t else sr[key_r[(x, y)]]
            ok = abs(m) <= eps_heat or abs(hv - thr) <= eps_heat or abs(float(sc) - cut) <= 10 * eps_score
            if not ok:
                unexplained.append(((x, y), hv, m, float(sc), cut))
        rep["unexplained"] = unexplained
        assert not unexplained, rep
        assert n_exc <= max(2, max_exceptions_frac * max(len(kr), 1)), rep
    else:
        assert len(kt) == len(kr), rep
    return rep


def compare_matches(m0_t, m1_t, m0_r, m1_r, oracle=None, eps=2e-6, max_exceptions_frac=0.005):
    """Match lists as COORDINATES: m0_* (n,2) key-points of image 0, m1_* (n,2) of image 1
    (what match_xfeat returns).  Coordinates, not indices, because the key-point order
    inside score-tie groups is free (finding 7).  The pair sets must be equal; a differing
    pair is accepted only when the oracle's similarity matrix shows a (near-)tie for the
    arg-max that decides it.  oracle = dict(kp0, kp1, d0, d1) of the oracle's key-points and
    descriptors for both images."""
    def pairs(a, b):
        a, b = _as_np(a), _as_np(b)
        return {(float(p[0]), float(p[1])): (float(q[0]), float(q[1])) for p, q in zip(a, b)}
    t, r = pairs(m0_t, m1_t), pairs(m0_r, m1_r)
    assert len(t) == len(_as_np(m0_t)), "duplicate image-0 key-point in test matches"
    diff = [k for k in set(t) | set(r) if t.get(k) != r.get(k)]
    rep = {"n_test": len(t), "n_ref": len(r), "differing_rows": len(diff)}
    if diff:
        assert oracle is not None, rep
        k0 = {(float(x), float(y)): i for i, (x, y) in enumerate(_as_np(oracle["kp0"]))}
        k1 = {(float(x), float(y)): i for i, (x, y) in enumerate(_as_np(oracle["kp1"]))}
        s = (oracle["d0"].double() @ oracle["d1"].double().t()).numpy()
        unexplained = []
        for k in diff:
            ok = False
            if k in k0:
                row = s[k0[k]]
                cands = {int(row.argmax())}
                for q in (t.get(k), r.get(k)):
                    if q is not None and q in k1:
                        cands.add(k1[q])
                rs = np.sort(row)[::-1]
                for j in cands:
                    cs = np.sort(s[:, j])[::-1]
                    if (rs[0] - rs[1]) <= eps or (cs[0] - cs[1]) <= eps:
                        ok = True
            if not ok:
                unexplained.append((k, t.get(k), r.get(k)))
        rep["unexplained"] = unexplained
        assert not unexplained, rep
        assert len(diff) <= max(2, max_exceptions_frac * max(len(r), 1)), rep
    return rep


def assert_close(a, b, tol, name=""):
    a, b = _as_np(a), _as_np(b)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    if a.size == 0:
        return 0.0
    d = float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max())
    assert d <= tol, f"{name}: max abs diff {d:.3e} > {tol:.1e}"
    return d


def compare_star_rows(test, ref, oracle=None, tol=2e-3, eps_sim=1e-4, eps_conf=5e-5, max_exceptions_frac=0.005,
                      test_dense=None):
    """match_xfeat_star rows (n,4) = (x0,y0 refined, x1,y1 = an image-1 dense key-point, never moved:
    xfeat.py:319-324).  Rows are keyed by (x1,y1) -- a mutual match uses every image-1 point at most once --
    and a test row equals a ref row when the refined (x0,y0) agree within `tol` pixels.

    Rows without a partner are exceptions.  With `oracle` = dict(sd, d0, d1, b) (the ORACLE's dense dicts and the
    batch index) each one must be explained by a deciding margin measured on the oracle's own numbers:
      * the mutual arg-max that (de)selects the pair is a near-tie of the raw similarity matrix (|gap| <= eps_sim;
        the entries are O(64), fp32 dot products of two correct evaluations differ by ~3e-5),
      * the confidence softmax(3*o).max() is within eps_conf of the 0.25 filter,
      * an end point is not in BOTH dense sets (`test_dense` = (kp0, kp1) of the tested path): a reliability top-k
        cut tie, reported by the dense-extraction comparison.
    Returns a report; raises on an unexplained row or on more than max_exceptions_frac exceptions."""
    t, r = _as_np(test).astype(np.float32), _as_np(ref).astype(np.float32)
    assert t.ndim == 2 and t.shape[1] == 4 and r.shape[1] == 4
    by_key = {}
    for i, row in enumerate(r):
        by_key.setdefault((float(row[2]), float(row[3])), []).append(i)
    used = np.zeros(len(r), bool)
    only_t, maxd = [], 0.0
    for row in t:
        cands = [i for i in by_key.get((float(row[2]), float(row[3])), []) if not used[i]]
        best = None
        for i in cands:
            d = float(np.abs(r[i, :2] - row[:2]).max())
            if d <= tol and (best is None or d < best[1]):
                best = (i, d)
        if best is None:
            only_t.append(row)
        else:
            used[best[0]] = True
            maxd = max(maxd, best[1])
    only_r = [r[i] for i in np.nonzero(~used)[0]]
    rep = {"n_test": len(t), "n_ref": len(r), "paired": int(used.sum()), "max_xy_diff": maxd,
           "only_test": len(only_t), "only_ref": len(only_r), "exceptions": len(only_t) + len(only_r)}
    if rep["exceptions"]:
        assert oracle is not None, rep
        from oracle import xfeat_oracle as O
        b = oracle["b"]
        f0, f1 = oracle["d0"]["descriptors"][b].double(), oracle["d1"]["descriptors"][b].double()
        k1 = oracle["d1"]["keypoints"][b].numpy()
        s = (f0 @ f1.t()).numpy()
        r12, r21 = s.argmax(1), s.argmax(0)
        kinds, unexplained = {"argmax_tie": 0, "conf_tie": 0, "topk_cut": 0}, []
        k1_index = {}
        for j, p in enumerate(k1):
            k1_index.setdefault((float(p[0]), float(p[1])), []).append(j)
        dense_sets = None
        if test_dense is not None:
            dense_sets = [set(map(lambda p: (float(p[0]), float(p[1])), _as_np(x))) for x in test_dense]
            ref_sets = [set(map(lambda p: (float(p[0]), float(p[1])), oracle[d]["keypoints"][b].numpy())) for d in ("d0", "d1")]
        for row in only_t + only_r:
            key = (float(row[2]), float(row[3]))
            ok = None
            if dense_sets is not None and (dense_sets[1] != ref_sets[1] or dense_sets[0] != ref_sets[0]):
                ok = "topk_cut"          # a different candidate set moves arg-maxes anywhere; the dense comparison bounds it
            for j in k1_index.get(key, []):
                i = int(r21[j])
                col, rw_ = np.sort(s[:, j])[::-1], np.sort(s[i])[::-1]
                if (col[0] - col[1]) <= eps_sim or (rw_[0] - rw_[1]) <= eps_sim:
                    ok = ok or "argmax_tie"
                if r12[i] == j:
                    o = O.fine_matcher(oracle["sd"], torch.cat([oracle["d0"]["descriptors"][b][i], oracle["d1"]["descriptors"][b][j]])[None])
                    conf = float(torch.softmax(o * 3, -1).max())
                    if abs(conf - 0.25) <= eps_conf:
                        ok = ok or "conf_tie"
            if ok:
                kinds[ok] += 1
            else:
                unexplained.append(row.tolist())
        rep["kinds"], rep["unexplained"] = kinds, unexplained
        assert not unexplained, rep
        assert rep["exceptions"] <= max(2, max_exceptions_frac * max(len(r), 1)), rep
    return rep


def compare_dense(test, ref, b, desc_tol=1e-4, eps_rel=2e-6, rel=None, max_exceptions_frac=0.005):
    """detectAndComputeDense outputs of batch item b: {'keypoints' (B,k,2), 'descriptors' (B,k,64), 'scales' (B,k)}.
    The result is the top-k of the reliability map per scale (xfeat.py:371), an unordered set as far as the matcher is
    concerned; the two scales can emit the same coordinate, so items are keyed by (x, y, scale).  Every common item's
    descriptor must agree within desc_tol (unconditionally); items on one side only are top-k cut ties and are bounded
    by max_exceptions_frac (with `rel` = the oracle's per-item reliability they must also sit within eps_rel of the cut)."""
    kt, kr = _as_np(test["keypoints"][b]), _as_np(ref["keypoints"][b])
    st, sr = _as_np(test["scales"][b]), _as_np(ref["scales"][b])
    dt, dr = _as_np(test["descriptors"][b]), _as_np(ref["descriptors"][b])
    assert kt.shape == kr.shape and dt.shape == dr.shape
    it = {(float(p[0]), float(p[1]), float(s)): i for i, (p, s) in enumerate(zip(kt, st))}
    ir = {(float(p[0]), float(p[1]), float(s)): i for i, (p, s) in enumerate(zip(kr, sr))}
    assert len(it) == len(kt), "duplicate (x,y,scale) in the tested dense set"
    common = [k for k in it if k in ir]
    a = np.array([it[k] for k in common]); c = np.array([ir[k] for k in common])
    rep = {"n": len(kt), "common": len(common), "only_test": len(it) - len(common), "only_ref": len(ir) - len(common),
           "desc_maxdiff": float(np.abs(dt[a] - dr[c]).max()) if len(common) else 0.0,
           "order_identical": bool(np.array_equal(kt, kr))}
    assert rep["desc_maxdiff"] <= desc_tol, rep
    assert rep["only_test"] <= max(2, max_exceptions_frac * len(kt)), rep
    return rep
