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
