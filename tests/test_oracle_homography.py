"""CPU checks of oracle/homography_oracle.py (the cv2.findHomography(USAC_MAGSAC) call of
/root/reference/realtime_demo.py:225).  cv2 is absent, so the restatement is pinned against the paper's definitions
(numerical integration of the marginalisation) and against synthetic ground truth."""
import numpy as np
import pytest
from scipy import integrate

from oracle import homography_oracle as ho


def synthetic_pair(n, outlier_frac, noise, seed, size=(640.0, 480.0)):
    """n correspondences under a random plausible homography: (p0, p1, H_true, inlier flags)."""
    g = np.random.default_rng(seed)
    w, h = size
    a = g.uniform(-0.35, 0.35)
    s = g.uniform(0.8, 1.25)
    H = np.array([[s * np.cos(a), -s * np.sin(a), g.uniform(-60, 60)],
                  [s * np.sin(a), s * np.cos(a), g.uniform(-40, 40)],
                  [g.uniform(-2e-4, 2e-4), g.uniform(-2e-4, 2e-4), 1.0]])
    p0 = np.stack([g.uniform(0, w, n), g.uniform(0, h, n)], axis=1)
    q = np.concatenate([p0, np.ones((n, 1))], axis=1) @ H.T
    p1 = q[:, :2] / q[:, 2:] + g.normal(0, noise, (n, 2))
    out = g.random(n) < outlier_frac
    p1[out] = np.stack([g.uniform(0, w, out.sum()), g.uniform(0, h, out.sum())], axis=1)
    return p0.astype(np.float32), p1.astype(np.float32), H, ~out


def transfer_error(H, Ht, size=(640.0, 480.0)):
    """max distance between the images of a 5x5 grid under the two homographies."""
    gx, gy = np.meshgrid(np.linspace(0, size[0], 5), np.linspace(0, size[1], 5))
    p = np.stack([gx.ravel(), gy.ravel(), np.ones(25)], axis=1)
    a, b = p @ H.T, p @ Ht.T
    return np.abs(a[:, :2] / a[:, 2:] - b[:, :2] / b[:, 2:]).max()


def test_weight_is_the_marginal_likelihood_of_the_paper():
    """w(r) = 1/sigma_max int_{r/k}^{sigma_max} g(r | sigma) d sigma with the trimmed chi density g (MAGSAC++ eq. 4-6), n = 4."""
    smax = 2 * 4.0 / ho.K_QUANTILE
    c = 0.25

    def g(sig, r):
        return 2 * c * sig ** -4 * r ** 3 * np.exp(-r * r / (2 * sig * sig))
    for r in (0.05, 0.7, 2.0, 5.0, 7.9):
        val, _ = integrate.quad(g, r / ho.K_QUANTILE, smax, args=(r,), epsabs=1e-13, epsrel=1e-11)
        assert abs(val / smax - ho.magsac_weight(r, smax)) < 1e-9
    assert ho.magsac_weight(8.0001, smax) == 0.0


def test_loss_is_the_integral_of_r_times_weight():
    """rho(r) = int_0^r x w(x) dx (the IRLS relation w = rho'(r) / r), constant beyond k sigma_max."""
    smax = 2 * 3.0 / ho.K_QUANTILE
    for r in (0.3, 1.5, 4.0, 5.9):
        val, _ = integrate.quad(lambda x: x * ho.magsac_weight(x, smax), 0, r, epsabs=1e-13, epsrel=1e-11)
        assert abs(val - ho.magsac_loss(r, smax)) < 1e-9
    assert ho.magsac_loss(6.0, smax) == ho.magsac_loss(50.0, smax)


def test_tables_are_monotone_fixed_point():
    bs, st, wt = ho.tables(4.0)
    assert bs == ho.NBINS / 64.0 and st.dtype == np.uint32 and st.shape == (ho.NBINS,)
    assert st[0] <= ho.SCORE_ONE and st[0] > 0.999 * ho.SCORE_ONE and st[-1] <= 2
    assert np.all(np.diff(st.astype(np.int64)) <= 0) and np.all(np.diff(wt) < 0)
    assert 0.999 < wt[0] <= 1.0 and wt[-1] >= 0


def test_samples_are_distinct_uniform_and_a_function_of_the_counter():
    idx, ok = ho.sample_sets(7, 3, 5000, 11)
    assert ok.all() and idx.min() == 0 and idx.max() == 10
    assert all(len(set(r)) == 4 for r in idx.tolist())
    cnt = np.bincount(idx.ravel(), minlength=11)
    assert cnt.min() > 0.85 * cnt.mean() and cnt.max() < 1.15 * cnt.mean()
    idx2, _ = ho.sample_sets(7, 3, 100, 11)
    assert np.array_equal(idx[:100], idx2)                      # hypothesis it does not depend on how many are drawn
    assert not np.array_equal(ho.sample_sets(8, 3, 100, 11)[0], idx2) and not np.array_equal(ho.sample_sets(7, 4, 100, 11)[0], idx2)
    idx4, ok4 = ho.sample_sets(1, 0, 300, 4)                    # n = 4: every sample is a permutation of all points
    assert ok4.mean() > 0.9 and all(sorted(r) == [0, 1, 2, 3] for r, o in zip(idx4.tolist(), ok4) if o)


def test_minimal_solver_interpolates_its_four_points():
    p0, p1, H, _ = synthetic_pair(64, 0.0, 0.0, 5)
    p0, p1 = p0.astype(np.float64), p1.astype(np.float64)
    idx, ok = ho.sample_sets(0, 0, 200, 64)
    h, valid = ho.minimal_homographies(p0, p1, idx)
    assert valid.mean() > 0.95
    r2 = ho.residuals_sq(h, p0, p1)
    for k in np.nonzero(valid)[0][:50]:
        assert r2[k, idx[k]].max() < 1e-12
        assert transfer_error(h[k] / h[k, 2, 2], H) < 2e-2       # the points are rounded to fp32: 3e-5 px lever
    # a sample whose image flips one triple is rejected; collinear samples too
    q1 = p1.copy()
    q1[idx[0, 3]] = 2 * q1[idx[0, 0]] - q1[idx[0, 3]] + (q1[idx[0, 1]] - q1[idx[0, 2]])
    _, v2 = ho.minimal_homographies(p0, q1, idx[:1])
    line = np.stack([np.arange(8.0), 2 * np.arange(8.0)], axis=1)
    _, v3 = ho.minimal_homographies(line, line, np.array([[0, 1, 2, 3]]))
    assert not v3[0]
    assert v2.dtype == bool


@pytest.mark.parametrize("n,outliers,noise", [(40, 0.0, 0.0), (300, 0.3, 0.5), (1500, 0.6, 1.0), (4096, 0.45, 0.7)])
def test_estimator_recovers_the_homography_and_its_inliers(n, outliers, noise):
    p0, p1, H, inl = synthetic_pair(n, outliers, noise, seed=n)
    Hh, mask, info = ho.find_homography(p0, p1, 4.0, max_iters=700, confidence=0.995, seed=1, return_info=True)
    assert info["found"] == 1 and Hh.shape == (3, 3) and Hh[2, 2] == 1.0 and mask.shape == (n, 1) and mask.dtype == np.uint8
    assert transfer_error(Hh, H) < (1e-3 if noise == 0 else 0.6)
    m = mask[:, 0] > 0
    q = np.concatenate([p0, np.ones((n, 1))], axis=1).astype(np.float64) @ H.T
    err_true = np.linalg.norm(q[:, :2] / q[:, 2:] - p1, axis=1)
    assert (m & (err_true > 6.0)).sum() == 0 and (~m & (err_true < 2.0)).sum() == 0
    assert abs(int(m.sum()) - int(inl.sum())) <= 0.03 * n + 2
    assert info["iters"] <= 700 and 0 <= info["best_it"] < info["iters"]
    if outliers == 0.0:
        assert info["iters"] < 20                              # all inliers: the confidence test stops the loop at once
    # refinement never lowers the model quality, and the result does not depend on anything but the arguments
    again = ho.find_homography(p0, p1, 4.0, max_iters=700, confidence=0.995, seed=1)
    assert np.array_equal(again[0], Hh) and np.array_equal(again[1], mask)


def test_refinement_improves_on_the_minimal_model():
    p0, p1, H, _ = synthetic_pair(800, 0.4, 1.0, seed=3)
    errs = []
    for lo in (0, ho.LO_ITERS):
        saved = ho.LO_ITERS
        ho.LO_ITERS = lo
        try:
            errs.append(transfer_error(ho.find_homography(p0, p1, 4.0, seed=5)[0], H))
        finally:
            ho.LO_ITERS = saved
    assert errs[1] < 0.5 * errs[0]


def test_degenerate_inputs():
    assert ho.find_homography(np.zeros((3, 2)), np.zeros((3, 2)), 4.0) == (None, None)
    assert ho.find_homography(np.zeros((0, 2)), np.zeros((0, 2)), 4.0) == (None, None)
    line = np.stack([np.arange(50.0), 3 * np.arange(50.0)], axis=1)
    assert ho.find_homography(line, line + 1, 4.0) == (None, None)              # all collinear: no valid sample
    g = np.random.default_rng(0)
    a, b = g.uniform(0, 500, (200, 2)), g.uniform(0, 500, (200, 2))
    H, mask = ho.find_homography(a, b, 1.0)                                       # no structure: nothing or a handful of chance inliers
    assert H is None or mask.sum() < 12
    sq = np.array([[0, 0], [100, 0], [100, 100], [0, 100]], np.float32)
    H, mask = ho.find_homography(sq, sq * 2 + 5, 4.0)                              # exactly four points
    assert mask.sum() == 4 and np.allclose(H, [[2, 0, 5], [0, 2, 5], [0, 0, 1]], atol=1e-9)


from sklearn.base import BaseEstimator, RegressorMixin  # noqa: E402


class _DltHomography(RegressorMixin, BaseEstimator):
    """Minimal scikit-learn estimator: X (n,2) -> y (n,2) under a homography fitted by the normalised DLT (SVD).  Shares no code with the oracle."""

    @staticmethod
    def _norm(p):
        c = p.mean(0)
        s = np.sqrt(2) / max(np.linalg.norm(p - c, axis=1).mean(), 1e-12)
        return np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])

    def fit(self, X, y):
        T0, T1 = self._norm(X), self._norm(y)
        a = np.c_[X, np.ones(len(X))] @ T0.T
        b = np.c_[y, np.ones(len(y))] @ T1.T
        rows = []
        for (x, yy, _), (u, v, _) in zip(a, b):
            rows.append([x, yy, 1, 0, 0, 0, -u * x, -u * yy, -u])
            rows.append([0, 0, 0, x, yy, 1, -v * x, -v * yy, -v])
        h = np.linalg.svd(np.asarray(rows))[2][-1].reshape(3, 3)
        self.H_ = np.linalg.inv(T1) @ h @ T0
        return self

    def predict(self, X):
        q = np.c_[X, np.ones(len(X))] @ self.H_.T
        return q[:, :2] / q[:, 2:]

    def score(self, X, y):
        return -float(np.mean(np.sum((self.predict(X) - y) ** 2, axis=1)))


@pytest.mark.parametrize("n,outliers,noise", [(400, 0.3, 0.5), (1500, 0.5, 1.0)])
def test_agrees_with_an_independent_ransac(n, outliers, noise):
    """scikit-learn's RANSACRegressor around a textbook DLT (different sampler, plain inlier counting, SVD instead of normal equations) finds the
    same model: inlier sets agree on >= 97 % of the correspondences and the two homographies move the image corners by < 0.5 px apart."""
    from sklearn.linear_model import RANSACRegressor
    p0, p1, _, _ = synthetic_pair(n, outliers, noise, seed=900 + n)
    H, mask = ho.find_homography(p0, p1, 4.0, seed=3)
    with np.errstate(all="ignore"):
        rs = RANSACRegressor(_DltHomography(), min_samples=4, residual_threshold=4.0, max_trials=700, random_state=0,
                             loss=lambda y, yp: np.sqrt(np.sum((y - yp) ** 2, axis=1)))
        rs.fit(p0.astype(np.float64), p1.astype(np.float64))
    agree = (rs.inlier_mask_ == (mask[:, 0] > 0)).mean()
    assert agree >= 0.97, agree
    assert transfer_error(H, rs.estimator_.H_ / rs.estimator_.H_[2, 2]) < 0.5
