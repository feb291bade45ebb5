"""Descriptor sets built to stress the fp16 filter window of xfh_match_mnn (csrc/k_match_f16.hip), and a numpy emulation of that filter.

The filter multiplies fp16 roundings of the descriptors; its window must contain the exact arg-max of every row and column whatever the
rounding errors do.  Random data keeps the errors ~20 sigma inside the window, so these sets ALIGN them: components a hair below / above
fp16 rounding midpoints, with the true best match losing in the rounded product and a competitor winning.  `emulate_filter` restates the
documented filter (scales, window, 32-wide blocks) in float64 on float16-rounded inputs so the CPU suite can check the window -- and that
half of it would NOT be enough on the same sets (the test has teeth) -- without a GPU.
"""
import numpy as np
import torch

U = 2.0 ** -11          # fp16 unit round-off (11-bit significand, round to nearest even)
C_WINDOW = 1.03 * 2.0 ** -10
KAPPA = 1.0e-5


def _case_b(n, seed, swap=False, unit=False):
    """n rows a_i (components +-1/8 with a random sign pattern per row, so that different rows are nearly orthogonal), each with a true best
    column t_i (supported where a_i rounds DOWN, itself rounding down: the fp16 product loses 2u) and a competitor h_i (supported where a_i
    rounds UP, itself rounding up: the product gains 2u) whose exact similarity is lower by ~2e-5.  D2 = [t_0 .. t_{n-1}, h_0 .. h_{n-1}]:
    with n >= 32 the two sit in different 32-wide blocks, and everything else in the block of t_i is ~0.1 -- the filter keeps that block only if
    its window covers a rounded-product deficit of 1.9u.  unit: 63 components, so that |a_i| < 1 (the caller-provided-copies path).
    swap: the roles of D1 and D2 exchanged (column side)."""
    rs = np.random.RandomState(seed)
    lo, hi = 1.0 + 0.999 * U, 1.0 + 1.0005 * U          # in units of 1/8: just below / above an fp16 rounding midpoint (any power-of-two scale)
    d1 = np.zeros((n, 64)); d2 = np.zeros((2 * n, 64))
    for i in range(n):
        perm = rs.permutation(64)
        sg = rs.choice([-1.0, 1.0], 64)
        down, up = (perm[:32], perm[32:63]) if unit else (perm[:32], perm[32:])
        d1[i, down] = lo * sg[down]; d1[i, up] = hi * sg[up]
        t = down[:len(up)]                                  # as many components as the competitor has
        d2[i, t] = lo * sg[t]                               # true best: S = k lo^2, rounded product k
        d2[n + i, up] = hi * sg[up]                         # competitor: rounded product (k-1) (1+2u)^2 + (1+2u)(1-2u)
        d2[n + i, up[-1]] = (1.0 - 2.0 * U) * sg[up[-1]]
    d1 /= 8.0; d2 /= 8.0
    d1, d2 = d1.astype(np.float32), d2.astype(np.float32)
    return (d2, d1) if swap else (d1, d2)


def _aligned_unit(n1, n2, seed):
    """Unit-norm-bounded rows (|row| <= 1) whose components all sit 0.49 ulp inside their fp16 grid point (in the scaled domain 256 x): every
    component of D1 and of half of D2 rounds AWAY from zero; the other half of D2 lies on the grid.  Genuine matches are planted."""
    rs = np.random.RandomState(seed)

    def rows(n, shrink_mask):
        v = rs.randn(n, 64)
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        q = (256.0 * (1.0 - 5e-4) * v).astype(np.float16).astype(np.float64)
        ulp = np.spacing(np.abs(q).astype(np.float16)).astype(np.float64)
        x = q - 0.49 * ulp * np.sign(q) * shrink_mask[:, None]
        return x / 256.0

    a = rows(n1, np.ones(n1))
    b = rows(n2, (np.arange(n2) % 2).astype(np.float64))
    m = min(n1, n2) // 2
    # planted matches: b_j = a_j re-quantised (on-grid copy: no rounding error on the b side, full error on the a side)
    b[:m:2] = (256.0 * a[:m:2]).astype(np.float16).astype(np.float64) / 256.0 * (1.0 - 3e-4)
    return a.astype(np.float32), b.astype(np.float32)


def sets():
    """(name, D1, D2, unit) -- unit: rows satisfy |row| <= 1.00001, so the caller-provided-fp16-copies path applies too."""
    out = []
    t = np.float32((16.0 + 0.4992 * 2.0 ** -6) / 256.0)         # 256 t just below the first midpoint of the [16, 32) binade: rounds down by ~u, the worst case
    out.append(("all_equal_round_down", np.full((40, 64), t, np.float32), np.full((70, 64), t, np.float32), True))
    for n, seed in ((64, 1), (200, 2)):
        d1, d2 = _case_b(n, seed)
        out.append((f"case_b_rows_{n}", d1, d2, False))
        e1, e2 = _case_b(n, seed + 10, swap=True)
        out.append((f"case_b_cols_{n}", e1, e2, False))
    d1, d2 = _case_b(96, 3, unit=True)
    out.append(("case_b_unit_rows_96", d1, d2, True))
    e1, e2 = _case_b(96, 4, swap=True, unit=True)
    out.append(("case_b_unit_cols_96", e1, e2, True))
    # the same with a dominating unrelated row on the other side (max norm 4x larger: the scale, and with it the absolute term, changes)
    d1, d2 = _case_b(64, 5)
    big = np.zeros((1, 64), np.float32); big[0, :2] = (2.0, -2.0)
    out.append(("case_b_with_big_row", np.concatenate([d1, big]), np.concatenate([d2, -big]), False))
    for n1, n2, seed in ((300, 500, 7), (1000, 64, 8), (2100, 2100, 9)):
        a, b = _aligned_unit(n1, n2, seed)
        out.append((f"aligned_unit_{n1}x{n2}", a, b, True))
    return out


def f16_scale(maxnorm):
    if not maxnorm > 0:
        return 1.0
    m, e = np.frexp(np.float32(maxnorm))
    return float(np.ldexp(1.0, 8 - int(e)))


def emulate_filter(d1, d2, c=C_WINDOW, kappa=KAPPA, unit=False):
    """The documented filter in numpy: returns (row_flags (n1, ceil(n2/32)), col_flags (n2, ceil(n1/32))) -- the 32-wide blocks the refine
    would visit.  Products of float16-rounded, scaled inputs summed in float64 (the hardware sums in fp32: covered by the window's slack)."""
    a, b = d1.astype(np.float32), d2.astype(np.float32)
    if unit:
        na = np.full(len(a), 1.00001); nb = np.full(len(b), 1.00001); ma = mb = 1.00001; sa = sb = 256.0
    else:
        na = np.sqrt((a.astype(np.float32) ** 2).sum(1, dtype=np.float32)) * np.float32(1.000001)
        nb = np.sqrt((b.astype(np.float32) ** 2).sum(1, dtype=np.float32)) * np.float32(1.000001)
        ma, mb = float(na.max()), float(nb.max())
        sa, sb = f16_scale(ma), f16_scale(mb)
    ah = (a * np.float32(sa)).astype(np.float16).astype(np.float64)
    bh = (b * np.float32(sb)).astype(np.float16).astype(np.float64)
    sh = ah @ bh.T
    ss = sa * sb
    e_row = ss * (c * na * mb + kappa * ma * mb)
    e_col = ss * (c * nb * ma + kappa * ma * mb)
    n1, n2 = sh.shape

    def blocks(mat, e):       # mat (n, m): block maxima over 32 columns against rowmax - 2e
        nb_ = (mat.shape[1] + 31) // 32
        pad = np.full((mat.shape[0], nb_ * 32), -np.inf); pad[:, :mat.shape[1]] = mat
        bm = pad.reshape(mat.shape[0], nb_, 32).max(2)
        return bm >= (mat.max(1) - 2.0 * e)[:, None]

    return blocks(sh, e_row), blocks(sh.T, e_col)


def exact_argmax_blocks(d1, d2):
    """Blocks holding the arg-max (and every near-tie within 1e-9) of the float64 similarity, per row and per column."""
    s = d1.astype(np.float64) @ d2.astype(np.float64).T

    def blocks(mat):
        nb_ = (mat.shape[1] + 31) // 32
        out = np.zeros((mat.shape[0], nb_), bool)
        mx = mat.max(1)
        ii, jj = np.nonzero(mat >= mx[:, None] - 1e-9 * np.maximum(1.0, np.abs(mx[:, None])))
        out[ii, jj // 32] = True
        return out

    return blocks(s), blocks(s.T)


def check_mnn_fp64(d1, d2, i0, i1, min_cossim=-1.0, rtol=2e-6):
    """A mutual-NN list against float64 similarities with the tie allowance of the parity contract: every reported pair is a row AND column
    maximum up to rtol, rows ascend, and every pair that wins its row and column by more than rtol is reported."""
    s = d1.astype(np.float64) @ d2.astype(np.float64).T
    tol = rtol * max(1.0, float(np.abs(s).max()))
    i0 = np.asarray(i0); i1 = np.asarray(i1)
    assert len(i0) == len(i1) and (len(i0) < 2 or (np.diff(i0) > 0).all())
    rmax, cmax = s.max(1), s.max(0)
    v = s[i0, i1]
    assert (v >= rmax[i0] - tol).all() and (v >= cmax[i1] - tol).all(), "a reported pair is not a mutual maximum"
    if min_cossim > 0:
        assert (v > min_cossim - tol).all()
    # strict winners: unique row maximum and unique column maximum by more than tol
    j = s.argmax(1)
    srt = np.sort(s, axis=1)
    row_strict = (srt[:, -1] - (srt[:, -2] if s.shape[1] > 1 else -np.inf)) > tol
    csrt = np.sort(s, axis=0)
    col_strict = (csrt[-1] - (csrt[-2] if s.shape[0] > 1 else -np.inf)) > tol
    must = [i for i in range(s.shape[0]) if row_strict[i] and col_strict[j[i]] and s[:, j[i]].argmax() == i and (min_cossim <= 0 or s[i, j[i]] > min_cossim + tol)]
    got = dict(zip(i0.tolist(), i1.tolist()))
    missing = [i for i in must if got.get(i) != int(j[i])]
    assert not missing, f"{len(missing)} strict mutual matches not reported, e.g. row {missing[:3]}"
    return len(must)


def as_torch(x):
    return torch.from_numpy(np.ascontiguousarray(x))
