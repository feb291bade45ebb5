"""CPU check of the fp16 filter window of xfh_match_mnn (csrc/k_match_f16.hip) on rounding-aligned adversarial descriptor sets
(tests/adversarial.py): the documented filter, restated in numpy, keeps the block of every exact arg-max -- and would not with half
the window, so the sets do discriminate.  The GPU suite runs the same sets through the kernels (test_gpu_parity.py)."""
import os
import re

import numpy as np
import pytest

import adversarial as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETS = A.sets()


def test_window_constants_are_the_kernel_s():
    src = open(os.path.join(ROOT, "accelerated_features_amd", "csrc", "k_match_f16.hip")).read()
    m = re.search(r"F16_C = ([0-9.]+)f \* ([0-9.]+)f;", src)
    assert m and abs(float(m.group(1)) * float(m.group(2)) - A.C_WINDOW) < 1e-12
    m = re.search(r"F16_KAPPA = ([0-9.e+-]+)f;", src)
    assert m and abs(float(m.group(1)) - A.KAPPA) < 1e-12
    # the derivation's terms: 2u + u^2 (rounding), 64 * 2^-23 (1+u)^2 (MFMA accumulation), 18 * 2^-24 (the refine's own dot product)
    u = 2.0 ** -11
    need = 2 * u + u * u + 64 * 2.0 ** -23 * (1 + u) ** 2 + 18 * 2.0 ** -24
    assert need < A.C_WINDOW and A.C_WINDOW < 1.05 * need
    assert 9 * 2.0 ** -20 < A.KAPPA                      # 9 (tau_a |b| + tau_b |a|) with tau <= 2^-21 maxnorm


@pytest.mark.parametrize("name,d1,d2,unit", SETS, ids=[s[0] for s in SETS])
def test_filter_keeps_every_exact_argmax(name, d1, d2, unit):
    need_r, need_c = A.exact_argmax_blocks(d1, d2)
    for um in ([False, True] if unit else [False]):
        fr, fc = A.emulate_filter(d1, d2, unit=um)
        assert not (need_r & ~fr).any() and not (need_c & ~fc).any(), (name, um)
    if unit:
        assert np.linalg.norm(d1, axis=1).max() <= 1.00001 and np.linalg.norm(d2, axis=1).max() <= 1.00001


def test_half_the_window_would_lose_argmaxes():
    """The sets have teeth: with c/2 (roughly the mistake of the round-2 bf16 window, which took u one bit too small) the filter drops the
    block of the true best match of every constructed row / column."""
    lost = 0
    for name, d1, d2, unit in SETS:
        if not name.startswith("case_b") or "big" in name:
            continue
        need_r, need_c = A.exact_argmax_blocks(d1, d2)
        fr, fc = A.emulate_filter(d1, d2, c=A.C_WINDOW / 2, kappa=A.KAPPA / 2)
        miss = int((need_r & ~fr).sum() + (need_c & ~fc).sum())
        assert miss >= min(len(d1), len(d2)), (name, miss)
        lost += miss
    assert lost > 500
    # caller-provided unit copies: the window is priced at |row| <= 1.00001 although these rows have norm 0.99 / 0.70, so it takes a third
    name, d1, d2, _ = [s for s in SETS if s[0] == "case_b_unit_rows_96"][0]
    need_r, need_c = A.exact_argmax_blocks(d1, d2)
    fr, fc = A.emulate_filter(d1, d2, c=A.C_WINDOW * 0.35, kappa=A.KAPPA * 0.35, unit=True)
    assert int((need_r & ~fr).sum()) >= 96


def test_measured_error_stays_inside_the_bound():
    """|S - S^| / E over every pair of every set: below 1 (the bound), and the aligned sets do come close to it (> 0.8)."""
    worst = 0.0
    for name, d1, d2, unit in SETS:
        a, b = d1.astype(np.float64), d2.astype(np.float64)
        na = np.linalg.norm(a, axis=1); nb = np.linalg.norm(b, axis=1)
        sa, sb = A.f16_scale(float(na.max() * 1.000001)), A.f16_scale(float(nb.max() * 1.000001))
        ah = (d1 * np.float32(sa)).astype(np.float16).astype(np.float64) / sa
        bh = (d2 * np.float32(sb)).astype(np.float16).astype(np.float64) / sb
        err = np.abs(a @ b.T - ah @ bh.T)
        e = A.C_WINDOW * np.outer(na, nb) + A.KAPPA * na.max() * nb.max()
        r = float((err / e).max())
        assert r < 1.0, (name, r)
        worst = max(worst, r)
    assert worst > 0.8, worst


def test_block_maxima_and_thresholds_round_the_conservative_way():
    """k_match_f16.hip stores block maxima and thresholds as fp16 (round 5): f16_up(x) = fp16(x + 2^-10 |x| + 2^-24) must never be below x, f16_down(x) never above it --
    then R >= thr in fp32 implies R16 >= thr16 and the flagged set only grows -- and neither may move a value by more than ~2^-9 of its size (the price: a few more blocks
    evaluated exactly).  The constants are parsed from the source; fp32 operations and the fp16 rounding are restated in numpy (round-to-nearest-even both)."""
    src = open(os.path.join(ROOT, "accelerated_features_amd", "csrc", "k_match_f16.hip")).read()
    m = re.search(r"f16_up\(float x\) \{ return \(_Float16\)\(__builtin_fmaf\(__builtin_fabsf\(x\), ([0-9.e+-]+)f, x\) \+ ([0-9.e+-]+)f\); \}", src)
    n = re.search(r"f16_down\(float x\) \{ return \(_Float16\)\(__builtin_fmaf\(__builtin_fabsf\(x\), -([0-9.e+-]+)f, x\) - ([0-9.e+-]+)f\); \}", src)
    assert m and n and m.groups() == n.groups()
    rel, tiny = np.float32(m.group(1)), np.float32(m.group(2))
    assert float(rel) == 2.0 ** -10 and float(tiny) == 2.0 ** -24
    scale = float(re.search(r"F16_STORE_SCALE = ([0-9.]+)f;", src).group(1))
    assert scale * 65536.0 * (1 + 2.0 ** -9) < 65504.0            # the largest scaled product (unit rows at scale 256 on both sides) stays finite in fp16
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.standard_normal(200000) * 10.0 ** rng.uniform(-9, 4.2, 200000), [0.0, -0.0, 16384.0, -16384.0, 2.0 ** -14, -2.0 ** -14, 2.0 ** -24, 6e-8, -6e-8, 1e-30, 65536.0 * scale],
                        np.float32(2.0) ** rng.integers(-24, 14, 1000) * (1 + 2.0 ** -11)]).astype(np.float32)      # (incl. fp16 subnormals, ties of the fp16 rounding, the top of the range)

    def fma32(a, b, c):      # one fp32 rounding of a * b + c (a * b is exact in fp64 for fp32 inputs, the sum to 2^-53: no double rounding that matters here is possible across 2^-10)
        return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(np.float32)
    up = (fma32(np.abs(x), rel, x) + tiny).astype(np.float32).astype(np.float16).astype(np.float64)
    down = (fma32(np.abs(x), -rel, x) - tiny).astype(np.float32).astype(np.float16).astype(np.float64)
    xd = x.astype(np.float64)
    assert (up >= xd).all() and (down <= xd).all()
    reach = np.abs(xd) * 2.0 ** -9 + 2.0 ** -22
    assert (up <= xd + reach).all() and (down >= xd - reach).all()
    # and through the comparison the refine makes (the sign of the fp16 difference): whenever R >= thr holds in fp32, the stored pair compares the same way
    r, t = x[:100000], x[100000:200000]
    keep = r >= t
    r16 = (fma32(np.abs(r), rel, r) + tiny).astype(np.float32).astype(np.float16)
    t16 = (fma32(np.abs(t), -rel, t) - tiny).astype(np.float32).astype(np.float16)
    d = (r16 - t16)                                                  # fp16 subtraction, as v_sub_f16 does it
    flagged = ~np.signbit(d)
    assert flagged[keep].all()
