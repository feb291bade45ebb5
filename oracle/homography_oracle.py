"""CPU oracle for the homography consumer of the matches.

TEST INFRASTRUCTURE ONLY.  Nothing under ``accelerated_features_amd/`` may import this module.

The call it stands for -- /root/reference/realtime_demo.py:223-229:

    self.H, inliers = cv2.findHomography(points1, points2, cv2.USAC_MAGSAC, self.ransac_thr, maxIters=700, confidence=0.995)
    inliers = inliers.flatten() > 0

PARITY UNPINNED: the arithmetic lives in OpenCV (opencv-contrib-python-headless==4.10.0.84,
/root/reference/requirements.txt:1), which is neither under /root/reference nor installed here, and the
reference holds no test or golden vector for this call.  What is restated is the PUBLISHED algorithm behind
``USAC_MAGSAC`` -- Barath, Noskova, Ivashechkin, Matas, "MAGSAC++, a fast, reliable and accurate robust
estimator", CVPR 2020 -- inside a plain RANSAC loop with the call site's arguments (threshold, maxIters,
confidence):

  * minimal sample = 4 correspondences, uniform, from a counter-based generator (so that hypothesis ``it`` is a
    function of (seed, pair, it) alone and a device can evaluate all of them at once while the stopping rule is
    applied afterwards exactly as a sequential loop would apply it);
  * 4-point homography from two projective-basis changes (3x3 adjugates; no pivoting); a sample is rejected when
    it does not keep the orientation of its four point triples (a real homography keeps or flips all of them);
  * model quality = MAGSAC++ marginalised loss (n = 4 degrees of freedom, k = 3.64, sigma_max = MAX_THR_FACTOR *
    threshold / k) of the forward transfer error, looked up -- as the library does -- in a table over the squared
    residual; the table holds 20-bit fixed-point values, so a score is an integer and no summation order exists;
  * standard RANSAC termination from the inlier ratio at ``threshold`` and ``confidence``;
  * sigma-consensus++ refinement of the winner: iteratively re-weighted least squares (Hartley-normalised
    inhomogeneous DLT) with the MAGSAC++ weights, a step kept only while it raises the model quality;
  * inlier mask: forward transfer error < threshold under the final model.

The constants an implementation is free to choose (table size, MAX_THR_FACTOR, the number of refinement steps,
the generator) are OURS, not OpenCV's: outputs are comparable with ``cv2.findHomography`` as estimates of the
same homography (tests/test_oracle_homography.py pins the loss / weight formulas against numerical integration of
the paper's marginalisation and the estimator against synthetic ground truth), not bit for bit.
"""
import math

import numpy as np
from scipy import special

K_QUANTILE = 3.64            # 0.99 quantile of the chi distribution with 4 degrees of freedom (MAGSAC++ section 3)
DOF = 4
MAX_THR_FACTOR = 2.0         # residuals up to MAX_THR_FACTOR * threshold still carry weight
NBINS = 4096                 # table bins over r^2 in [0, t_max^2)
SCORE_ONE = 1 << 20          # fixed-point 1.0 of a table entry
LO_ITERS = 5                 # re-weighted least-squares steps on the winner
MAX_DRAWS = 16               # generator draws per sample before it is given up
MASK64 = (1 << 64) - 1


# --------------------------------------------------------------------------------------
# MAGSAC++ loss / weight (paper eq. 4-7 for n = 4)
# --------------------------------------------------------------------------------------
def _upper_gamma(a, x):
    return special.gammaincc(a, x) * special.gamma(a)


def _lower_gamma(a, x):
    return special.gammainc(a, x) * special.gamma(a)


def magsac_weight(r, sigma_max):
    """w(r) = 1/sigma_max * C(n) 2^((n-1)/2) [Gamma((n-1)/2, r^2 / 2 sigma_max^2) - Gamma((n-1)/2, k^2/2)], 0 beyond k sigma_max."""
    r = np.asarray(r, np.float64)
    c = 1.0 / (2.0 ** (DOF / 2.0) * special.gamma(DOF / 2.0))
    a = (DOF - 1) / 2.0
    w = c * 2.0 ** a / sigma_max * (_upper_gamma(a, r * r / (2 * sigma_max ** 2)) - _upper_gamma(a, K_QUANTILE ** 2 / 2))
    return np.where(r < K_QUANTILE * sigma_max, w, 0.0)


def magsac_loss(r, sigma_max):
    """rho(r) = int_0^r x w(x) dx in closed form; constant rho(k sigma_max) beyond k sigma_max."""
    r = np.minimum(np.asarray(r, np.float64), K_QUANTILE * sigma_max)
    c = 1.0 / (2.0 ** (DOF / 2.0) * special.gamma(DOF / 2.0))
    x = r * r / (2 * sigma_max ** 2)
    a = (DOF - 1) / 2.0
    return c * 2.0 ** ((DOF + 1) / 2.0) / sigma_max * (
        sigma_max ** 2 / 2 * _lower_gamma((DOF + 1) / 2.0, x) + r * r / 4 * (_upper_gamma(a, x) - _upper_gamma(a, K_QUANTILE ** 2 / 2)))


def tables(thr):
    """(bin_scale, score_table uint32[NBINS], weight_table float64[NBINS]) for threshold ``thr``; entries at the bin centres.
    score = 1 - rho(r)/rho(k sigma_max) in 20-bit fixed point, weight = w(r)/w(0)."""
    t_max = MAX_THR_FACTOR * float(thr)
    sigma_max = t_max / K_QUANTILE
    bin_scale = NBINS / (t_max * t_max)
    r = np.sqrt((np.arange(NBINS, dtype=np.float64) + 0.5) / bin_scale)
    q = 1.0 - magsac_loss(r, sigma_max) / magsac_loss(K_QUANTILE * sigma_max, sigma_max)
    score = np.floor(np.clip(q, 0.0, 1.0) * SCORE_ONE + 0.5).astype(np.uint32)
    weight = magsac_weight(r, sigma_max) / magsac_weight(0.0, sigma_max)
    return bin_scale, score, weight


# --------------------------------------------------------------------------------------
# sampling
# --------------------------------------------------------------------------------------
def _mix64(z):
    z = z.astype(np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
    return z ^ (z >> np.uint64(31))


def draw_index(seed, pair, it, draw, n):
    """Draw ``draw`` of hypothesis ``it``: splitmix64 finaliser of a counter, upper 32 bits scaled to [0, n)."""
    counter = (np.uint64(pair) * np.uint64(1 << 20) + np.asarray(it, np.uint64)) * np.uint64(MAX_DRAWS) + np.uint64(draw)
    h = _mix64(np.uint64(seed & MASK64) + np.uint64(0x9e3779b97f4a7c15) * (counter + np.uint64(1)))
    return (((h >> np.uint64(32)) * np.uint64(n)) >> np.uint64(32)).astype(np.int64)


def sample_sets(seed, pair, iters, n):
    """(idx (iters,4), ok (iters,)): four distinct indices per hypothesis, draws consumed in order, duplicates redrawn."""
    it = np.arange(iters, dtype=np.uint64)
    idx = np.full((iters, 4), -1, np.int64)
    slot = np.zeros(iters, np.int64)
    for d in range(MAX_DRAWS):
        cand = draw_index(seed, pair, it, d, n)
        dup = np.zeros(iters, bool)
        for s in range(4):
            dup |= (s < slot) & (idx[:, s] == cand)
        take = (slot < 4) & ~dup
        rows = np.nonzero(take)[0]
        idx[rows, slot[rows]] = cand[rows]
        slot[rows] += 1
    return idx, slot == 4


# --------------------------------------------------------------------------------------
# minimal solver, residuals (the operation order below IS the contract: every product and sum is rounded once)
# --------------------------------------------------------------------------------------
def _orient(ax, ay, bx, by, cx, cy):
    return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax)


def _basis(x, y):
    """Columns lambda_i * (x_i, y_i, 1), i = 0..2, with (lambda_0 p_0 + lambda_1 p_1 + lambda_2 p_2) = D p_3; also the four triple orientations."""
    l0 = _orient(x[:, 3], y[:, 3], x[:, 1], y[:, 1], x[:, 2], y[:, 2])
    l1 = _orient(x[:, 0], y[:, 0], x[:, 3], y[:, 3], x[:, 2], y[:, 2])
    l2 = _orient(x[:, 0], y[:, 0], x[:, 1], y[:, 1], x[:, 3], y[:, 3])
    d = _orient(x[:, 0], y[:, 0], x[:, 1], y[:, 1], x[:, 2], y[:, 2])
    lam = (l0, l1, l2)
    m = np.empty((x.shape[0], 3, 3))
    for j in range(3):
        m[:, 0, j] = lam[j] * x[:, j]
        m[:, 1, j] = lam[j] * y[:, j]
        m[:, 2, j] = lam[j]
    return m, (d, l0, l1, l2)


def _adjugate(a):
    c = np.empty_like(a)
    c[:, 0, 0] = a[:, 1, 1] * a[:, 2, 2] - a[:, 1, 2] * a[:, 2, 1]
    c[:, 0, 1] = a[:, 0, 2] * a[:, 2, 1] - a[:, 0, 1] * a[:, 2, 2]
    c[:, 0, 2] = a[:, 0, 1] * a[:, 1, 2] - a[:, 0, 2] * a[:, 1, 1]
    c[:, 1, 0] = a[:, 1, 2] * a[:, 2, 0] - a[:, 1, 0] * a[:, 2, 2]
    c[:, 1, 1] = a[:, 0, 0] * a[:, 2, 2] - a[:, 0, 2] * a[:, 2, 0]
    c[:, 1, 2] = a[:, 0, 2] * a[:, 1, 0] - a[:, 0, 0] * a[:, 1, 2]
    c[:, 2, 0] = a[:, 1, 0] * a[:, 2, 1] - a[:, 1, 1] * a[:, 2, 0]
    c[:, 2, 1] = a[:, 0, 1] * a[:, 2, 0] - a[:, 0, 0] * a[:, 2, 1]
    c[:, 2, 2] = a[:, 0, 0] * a[:, 1, 1] - a[:, 0, 1] * a[:, 1, 0]
    return c


def minimal_homographies(p0, p1, idx):
    """H (K,3,3) (arbitrary scale) through the four correspondences of each row of ``idx``; valid (K,)."""
    x0, y0 = p0[idx, 0], p0[idx, 1]
    x1, y1 = p1[idx, 0], p1[idx, 1]
    a, da = _basis(x0, y0)
    b, db = _basis(x1, y1)
    adj = _adjugate(a)
    h = np.empty_like(a)
    for i in range(3):
        for j in range(3):
            h[:, i, j] = (b[:, i, 0] * adj[:, 0, j] + b[:, i, 1] * adj[:, 1, j]) + b[:, i, 2] * adj[:, 2, j]
    prod = [da[t] * db[t] for t in range(4)]
    pos = (prod[0] > 0) & (prod[1] > 0) & (prod[2] > 0) & (prod[3] > 0)
    neg = (prod[0] < 0) & (prod[1] < 0) & (prod[2] < 0) & (prod[3] < 0)
    return h, pos | neg


def residuals_sq(h, p0, p1):
    """Squared forward transfer error |p1 - proj(H p0)|^2, (K,n) for H (K,3,3)."""
    x, y = p0[None, :, 0], p0[None, :, 1]
    w = (h[:, 2, 0, None] * x + h[:, 2, 1, None] * y) + h[:, 2, 2, None]
    u = (h[:, 0, 0, None] * x + h[:, 0, 1, None] * y) + h[:, 0, 2, None]
    v = (h[:, 1, 0, None] * x + h[:, 1, 1, None] * y) + h[:, 1, 2, None]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        iw = 1.0 / w
        dx = p1[None, :, 0] - u * iw
        dy = p1[None, :, 1] - v * iw
        return dx * dx + dy * dy


def quality(r2, thr, bin_scale, score_table):
    """(integer MAGSAC++ score, inlier count at thr) per row of r2 (K,n)."""
    t_max = MAX_THR_FACTOR * thr
    with np.errstate(invalid="ignore"):
        near = r2 < t_max * t_max
        b = np.where(near, r2 * bin_scale, 0.0).astype(np.int64)
        b = np.minimum(b, NBINS - 1)
        score = np.where(near, score_table[b].astype(np.int64), 0).sum(axis=1)
        cnt = (r2 < thr * thr).sum(axis=1)
    return score, cnt


def iterations_needed(inliers, n, confidence, max_iters):
    w = inliers / n
    p = 1.0 - w * w * w * w
    if p <= 0.0:
        return 1
    if p >= 1.0:
        return max_iters
    k = math.log(1.0 - confidence) / math.log(p)
    return int(min(float(max_iters), math.ceil(k)))


# --------------------------------------------------------------------------------------
# weighted least squares (sigma-consensus++ step)
# --------------------------------------------------------------------------------------
def _normalisation(p):
    c = p.mean(axis=0)
    d = np.sqrt(((p - c) ** 2).sum(axis=1)).mean()
    s = math.sqrt(2.0) / d if d > 0 else 1.0
    return c, s


def weighted_dlt(p0, p1, w, norm0, norm1):
    """argmin sum w_i |algebraic error|^2 with h33 = 1 in the normalised frames; None when the normal matrix is not positive definite."""
    (c0, s0), (c1, s1) = norm0, norm1
    x, y = (p0[:, 0] - c0[0]) * s0, (p0[:, 1] - c0[1]) * s0
    u, v = (p1[:, 0] - c1[0]) * s1, (p1[:, 1] - c1[1]) * s1
    z, o = np.zeros_like(x), np.ones_like(x)
    au = np.stack([x, y, o, z, z, z, -u * x, -u * y], axis=1)
    av = np.stack([z, z, z, x, y, o, -v * x, -v * y], axis=1)
    a = np.concatenate([au, av])
    ww = np.concatenate([w, w])
    rhs = np.concatenate([u, v])
    m = (a * ww[:, None]).T @ a
    g = (a * ww[:, None]).T @ rhs
    try:
        l = np.linalg.cholesky(m)
    except np.linalg.LinAlgError:
        return None
    hv = np.linalg.solve(l.T, np.linalg.solve(l, g))
    hn = np.append(hv, 1.0).reshape(3, 3)
    t0 = np.array([[s0, 0, -s0 * c0[0]], [0, s0, -s0 * c0[1]], [0, 0, 1.0]])
    t1inv = np.array([[1 / s1, 0, c1[0]], [0, 1 / s1, c1[1]], [0, 0, 1.0]])
    return t1inv @ hn @ t0


# --------------------------------------------------------------------------------------
# the estimator
# --------------------------------------------------------------------------------------
def find_homography(points1, points2, ransac_thr, max_iters=700, confidence=0.995, seed=0, pair=0, return_info=False):
    """Restatement of ``cv2.findHomography(points1, points2, cv2.USAC_MAGSAC, ransac_thr, maxIters=, confidence=)`` as used at
    /root/reference/realtime_demo.py:225.  Returns (H (3,3) float64 with H[2,2] = 1 or None, inliers (n,1) uint8 or None)."""
    p0 = np.asarray(points1, np.float32).reshape(-1, 2).astype(np.float64)
    p1 = np.asarray(points2, np.float32).reshape(-1, 2).astype(np.float64)
    n = p0.shape[0]
    info = {"found": 0, "best_it": -1, "iters": 0, "n_inliers": 0, "score": 0, "lo_accepted": 0}
    if n < 4 or p1.shape[0] != n:
        return (None, None, info) if return_info else (None, None)
    thr = float(ransac_thr)
    bin_scale, stab, wtab = tables(thr)
    idx, ok = sample_sets(seed, pair, max_iters, n)
    idx = np.where(ok[:, None], idx, 0)
    with np.errstate(invalid="ignore", over="ignore"):
        hyp, valid = minimal_homographies(p0, p1, idx)
    valid &= ok
    score = np.zeros(max_iters, np.int64)
    cnt = np.zeros(max_iters, np.int64)
    for a in range(0, n, 2048):                       # chunks of points: integer scores add up in any order
        s, c = quality(residuals_sq(hyp, p0[a:a + 2048], p1[a:a + 2048]), thr, bin_scale, stab)
        score += s
        cnt += c
    score[~valid] = 0
    # the stopping rule, applied as the sequential loop applies it
    best, best_s, k_stop, it = -1, 0, max_iters, 0
    while it < max_iters and it < k_stop:
        if score[it] > best_s:
            best, best_s = it, int(score[it])
            k_stop = min(k_stop, iterations_needed(int(cnt[it]), n, confidence, max_iters))
        it += 1
    info["iters"] = it
    if best < 0:
        return (None, None, info) if return_info else (None, None)
    # sigma-consensus++ on the winner
    norm0, norm1 = _normalisation(p0), _normalisation(p1)
    t_max2 = (MAX_THR_FACTOR * thr) ** 2
    h_best, s_best = None, 0
    h_cur = hyp[best]
    for step in range(LO_ITERS + 1):
        r2 = residuals_sq(h_cur[None], p0, p1)
        s = int(quality(r2, thr, bin_scale, stab)[0][0])
        if s <= s_best:
            break
        h_best, s_best = h_cur, s
        info["lo_accepted"] = step
        if step == LO_ITERS:
            break
        with np.errstate(invalid="ignore"):
            near = r2[0] < t_max2
            b = np.minimum(np.where(near, r2[0] * bin_scale, 0.0).astype(np.int64), NBINS - 1)
        w = np.where(near, wtab[b], 0.0)
        h_new = weighted_dlt(p0, p1, w, norm0, norm1)
        if h_new is None or not np.all(np.isfinite(h_new)):
            break
        h_cur = h_new
    r2 = residuals_sq(h_best[None], p0, p1)[0]
    with np.errstate(invalid="ignore"):
        mask = (r2 < thr * thr)
    n_in = int(mask.sum())
    info.update(best_it=best, n_inliers=n_in, score=s_best)
    if n_in < 4:
        return (None, None, info) if return_info else (None, None)
    info["found"] = 1
    h = h_best / h_best[2, 2] if abs(h_best[2, 2]) > 1e-300 else h_best / np.linalg.norm(h_best)
    out = (h, mask.astype(np.uint8).reshape(-1, 1))
    return out + (info,) if return_info else out
