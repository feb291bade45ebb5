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
