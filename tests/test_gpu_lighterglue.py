"""GPU parity of the LighterGlue HIP path (xfh_lg_match through the C ABI) against oracle/lighterglue_oracle.py.

The oracle is pinned block by block against the HuggingFace port (tests/test_oracle_lighterglue.py) and is
*unpinned* against kornia 0.7.2 itself (absent) -- see its header.  Bars: match pairs identical up to near-ties of the
assignment scores (tie-aware), matching scores within 1e-4, final descriptors within 2e-4 (six fp32 layers).
"""
import ctypes as C

import numpy as np
import pytest
import torch

import fixtures
from oracle import lighterglue_oracle as LG

pytestmark = pytest.mark.gpu
NO_PRUNING = 1 << 30


@pytest.fixture(scope="module")
def sd():
    return fixtures.lighterglue_state_dict(0)


@pytest.fixture(scope="module")
def lg(sd):
    from accelerated_features_amd.lighterglue import LighterGlue
    m = LighterGlue(weights=sd)
    assert m.dev.type == "cuda"
    return m


def _run(lg, inp, min_conf, prune_min_kpts):
    k0, d0, s0, k1, d1, s1 = inp
    m, s, c = lg.match_device(k0.cuda(), d0.cuda(), s0.tolist(), k1.cuda(), d1.cuda(), s1.tolist(), min_conf, prune_min_kpts)
    n = int(c.item())
    return m[:n].cpu().numpy(), s[:n].cpu().numpy()


def _compare(got_m, got_s, sd, inp, min_conf, prune, prune_min_kpts=-1, tol=1e-4):
    trace = []
    ref_m, ref_s = LG.lighterglue_forward(sd, *inp, min_conf=min_conf, prune=prune, prune_min_kpts=prune_min_kpts, trace=trace)
    ref_m, ref_s = ref_m.numpy(), ref_s.numpy()
    assert len(ref_m) >= 3, "test inputs must produce matches"
    assert np.all(np.diff(got_m[:, 0]) > 0), "matches must be ascending in image-0 index"
    ref = {(int(a), int(b)): float(s) for (a, b), s in zip(ref_m, ref_s)}
    got = {(int(a), int(b)): float(s) for (a, b), s in zip(got_m, got_s)}
    for key in set(ref) & set(got):
        assert abs(ref[key] - got[key]) <= tol * max(1.0, abs(ref[key])), (key, ref[key], got[key])
    # tie-aware: a pair present on one side only must be a near-tie of the oracle's assignment or sit on the threshold
    scores, _, ind0, ind1 = trace[-1]
    core = scores[:-1, :-1]
    pos0 = {int(v): i for i, v in enumerate(ind0.tolist())}
    pos1 = {int(v): i for i, v in enumerate(ind1.tolist())}
    for a, b in set(ref) ^ set(got):
        assert a in pos0 and b in pos1, f"pair ({a},{b}) uses a key-point the oracle pruned"
        i, j = pos0[a], pos1[b]
        v = float(core[i, j])
        borderline = abs(np.exp(v) - min_conf) < 1e-3
        if (a, b) in got:        # the oracle rejected it: either just under the threshold or not quite the mutual maximum
            near_tie = v >= float(core[i].max()) - 1e-3 and v >= float(core[:, j].max()) - 1e-3
        else:                    # the oracle's mutual maximum lost on the GPU: a runner-up must be that close
            row, col = core[i].clone(), core[:, j].clone()
            row[j], col[i] = -np.inf, -np.inf
            near_tie = max(float(row.max()), float(col.max())) >= v - 1e-3
        assert near_tie or borderline, (a, b, v)
    assert len(set(ref) ^ set(got)) <= max(1, len(ref) // 100), (len(ref), len(got), len(set(ref) ^ set(got)))
    return trace


@pytest.mark.parametrize("n0,n1", [(300, 257), (64, 1000), (1, 40), (513, 512)])
def test_lighterglue_without_pruning_vs_oracle(lg, sd, n0, n1):
    inp = fixtures.lighterglue_inputs(n0, n1, seed=n0)
    if min(n0, n1) < 8:                       # too small for the ">8 matches" sanity: only compare the lists
        got_m, got_s = _run(lg, inp, 0.0, NO_PRUNING)
        ref_m, ref_s = LG.lighterglue_forward(sd, *inp, min_conf=0.0, prune=False)
        assert np.array_equal(got_m, ref_m.numpy())
        np.testing.assert_allclose(got_s, ref_s.numpy(), rtol=1e-4, atol=1e-6)
        return
    got_m, got_s = _run(lg, inp, 0.01, NO_PRUNING)
    trace = _compare(got_m, got_s, sd, inp, 0.01, prune=False)
    # final descriptors of image 0 live at the start of the caller's workspace when nothing was pruned
    x = lg._ws[(-lg._ws.data_ptr()) % 256:][: n0 * 192 * 4].clone().view(torch.float32).reshape(n0, 192).cpu()
    ref_d0 = trace[-2][0]
    assert float((x[:, :96] - ref_d0).abs().max()) <= 2e-4 * max(1.0, float(ref_d0.abs().max()))


@pytest.mark.parametrize("min_kpts", [-1, 200])
def test_lighterglue_with_width_pruning_vs_oracle(lg, sd, min_kpts):
    inp = fixtures.lighterglue_inputs(700, 640, seed=3, size0=(640, 480), size1=(800, 600))
    got_m, got_s = _run(lg, inp, 0.01, min_kpts)
    trace = _compare(got_m, got_s, sd, inp, 0.01, prune=True, prune_min_kpts=min_kpts)
    sizes = [t[0].shape[0] for t in trace[:-1]]
    assert sizes[-1] < 700, "the fixture weights must prune something, otherwise this test checks nothing"


def test_lighterglue_full_size_4096_vs_oracle(lg, sd):
    """BASELINE size (top_k = 4096 key-points per image), the pruning threshold the class uses on a GPU (1536)."""
    inp = fixtures.lighterglue_inputs(4096, 4096, seed=9)
    got_m, got_s = _run(lg, inp, 0.01, 1536)
    _compare(got_m, got_s, sd, inp, 0.01, prune=True, prune_min_kpts=1536, tol=2e-4)


def test_lighterglue_8192_properties(lg):
    """Beyond the oracle's comfortable size: one-to-one, ascending, thresholded, bit-identical when repeated."""
    inp = fixtures.lighterglue_inputs(8192, 6000, seed=10)
    m1, s1 = _run(lg, inp, 0.02, 1536)
    m2, s2 = _run(lg, inp, 0.02, 1536)
    assert np.array_equal(m1, m2) and np.array_equal(s1, s2)
    assert len(m1) > 0 and np.all(np.diff(m1[:, 0]) > 0)
    assert len(np.unique(m1[:, 1])) == len(m1) and m1[:, 0].max() < 8192 and m1[:, 1].max() < 6000 and m1.min() >= 0
    assert float(s1.min()) > 0.02 and float(s1.max()) <= 1.0 + 1e-6


def test_lighterglue_class_surface_and_determinism(lg, sd):
    from accelerated_features_amd import XFeat
    inp = fixtures.lighterglue_inputs(400, 380, seed=5)
    k0, d0, s0, k1, d1, s1 = inp
    data = {'keypoints0': k0[None].cuda(), 'keypoints1': k1[None].cuda(), 'descriptors0': d0[None].cuda(), 'descriptors1': d1[None].cuda(),
            'image_size0': s0[None].cuda(), 'image_size1': s1[None].cuda()}
    out = lg(data, min_conf=0.05)
    out2 = lg(data, min_conf=0.05)
    assert torch.equal(out['matches'][0], out2['matches'][0]) and torch.equal(out['scores'][0], out2['scores'][0])
    assert out['matches'][0].dtype == torch.int64 and out['matches'][0].shape[1] == 2
    assert float(out['scores'][0].min()) > 0.05
    mm = out['matches'][0]
    assert torch.equal(out['matches0'][0, mm[:, 0]], mm[:, 1]) and torch.equal(out['matches1'][0, mm[:, 1]], mm[:, 0])
    assert int((out['matches0'] >= 0).sum()) == len(mm) and torch.equal(out['matching_scores0'][0, mm[:, 0]], out['scores'][0])
    # the XFeat-level wrapper (modules/xfeat.py:131-162): numpy (S,2), (S,2), (S,2)
    xf = XFeat(weights=fixtures.synthetic_state_dict(0))
    xf.lighterglue = lg
    a, b, idx = xf.match_lighterglue({'keypoints': k0.cuda(), 'descriptors': d0.cuda(), 'image_size': (640, 480)},
                                     {'keypoints': k1.cuda(), 'descriptors': d1.cuda(), 'image_size': (640, 480)}, min_conf=0.05)
    assert isinstance(a, np.ndarray) and a.shape == b.shape == idx.shape and idx.shape[1] == 2
    assert np.array_equal(idx, out['matches'][0].cpu().numpy())
    assert np.array_equal(a, k0.numpy()[idx[:, 0]]) and np.array_equal(b, k1.numpy()[idx[:, 1]])


def test_lighterglue_pair_batch_with_device_counts_equals_single_pairs(lg):
    """xfh_lg_match_pairs on zero-padded fixed-capacity lists == xfh_lg_match on the ragged lists."""
    cap, sizes = 512, [(500, 512), (37, 300), (512, 0), (130, 131)]
    kp = torch.zeros(2 * len(sizes), cap, 2)
    de = torch.zeros(2 * len(sizes), cap, 64)
    cnt = torch.zeros(2 * len(sizes), dtype=torch.int32)
    singles = []
    for p, (n0, n1) in enumerate(sizes):
        k0, d0, s0, k1, d1, s1 = fixtures.lighterglue_inputs(max(n0, 1), max(n1, 1), seed=40 + p)
        kp[2 * p, :n0], de[2 * p, :n0], kp[2 * p + 1, :n1], de[2 * p + 1, :n1] = k0[:n0], d0[:n0], k1[:n1], d1[:n1]
        cnt[2 * p], cnt[2 * p + 1] = n0, n1
        if n0 and n1:
            singles.append(_run(lg, (k0, d0, s0, k1, d1, s1), 0.02, -1))
        else:
            singles.append((np.zeros((0, 2), np.int64), np.zeros((0,), np.float32)))
    m, s, n = lg.match_pairs_device(kp.cuda(), de.cuda(), cnt.cuda(), (640, 480), 0.02, -1)
    n = n.cpu().tolist()
    for p in range(len(sizes)):
        assert n[p] == len(singles[p][0]), (p, n[p], len(singles[p][0]))
        assert np.array_equal(m[p, :n[p]].cpu().numpy(), singles[p][0])
        np.testing.assert_allclose(s[p, :n[p]].cpu().numpy(), singles[p][1], rtol=1e-4, atol=1e-7)      # the key-split count follows the capacity: summation order differs


def test_lighterglue_rejects_bad_arguments(lg):
    from accelerated_features_amd import _lib as L
    lib = L.load()
    with pytest.raises(L.XFeatHipError):
        L.check(lib.xfh_lg_match(lg.handle(), None, None, 10, 640.0, 480.0, None, None, 10, 640.0, 480.0, 0.1, -1, None, None, None, None, 0, None), "xfh_lg_match")
    assert lib.xfh_lg_workspace_bytes(0, 5) == 0
    with pytest.raises(RuntimeError):
        lg.load_state_dict({"input_proj.weight": torch.zeros(3, 3)})
