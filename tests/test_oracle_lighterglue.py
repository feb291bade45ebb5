"""LighterGlue oracle against golden vectors produced by the HuggingFace port of LightGlue with tied
weights (tests/golden/make_golden_lighterglue.py).  kornia 0.7.2 -- the reference's real dependency -- is absent:
these tests pin the building blocks to an independent implementation of the same published algorithm, not to kornia."""
import os

import numpy as np
import torch

import fixtures
from oracle import lighterglue_oracle as lg

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SD = fixtures.lighterglue_state_dict(0)


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_state_dict_layout_matches_reference_loader():
    keys = [k for k, _ in lg.state_dict_keys()]
    assert len(keys) == len(set(keys)) == 3 + 6 * 22 + 6 * 4 + 5 * 2
    # names the reference's loader produces (modules/lighterglue.py:41-46) from the checkpoint's self_attn.{i} / cross_attn.{i}
    assert "transformers.0.self_attn.Wqkv.weight" in keys and "transformers.5.cross_attn.to_qk.weight" in keys
    assert all(tuple(SD[k].shape) == s for k, s in lg.state_dict_keys())


def test_normalisation_and_positional_encoding_match_hf_port():
    g = np.load(os.path.join(G, "lg_posenc.npz"))
    for s in ("0", "1"):
        kn = lg.normalize_keypoints(_t(g["k" + s]), _t(g["size" + s]))
        assert torch.allclose(kn, _t(g["kn" + s]), atol=1e-6)
        cos, sin = lg.posenc(SD, kn)
        assert torch.allclose(cos, _t(g["cos" + s]), atol=2e-6) and torch.allclose(sin, _t(g["sin" + s]), atol=2e-6)


def test_transformer_layer_matches_hf_port():
    g, p = np.load(os.path.join(G, "lg_layer.npz")), np.load(os.path.join(G, "lg_posenc.npz"))
    e0, e1 = (_t(p["cos0"]), _t(p["sin0"])), (_t(p["cos1"]), _t(p["sin1"]))
    for i in (0, 3):
        y0, y1 = lg.transformer_layer(SD, i, _t(g["x0"]), _t(g["x1"]), e0, e1)
        assert torch.allclose(y0, _t(g[f"y0_{i}"]), atol=2e-5), float((y0 - _t(g[f"y0_{i}"])).abs().max())
        assert torch.allclose(y1, _t(g[f"y1_{i}"]), atol=2e-5)


def test_assignment_and_filter_match_hf_port():
    g = np.load(os.path.join(G, "lg_assign.npz"))
    d0, d1 = _t(g["d0"]), _t(g["d1"])
    scores = lg.log_assignment(SD, 5, d0, d1)
    assert torch.allclose(scores, _t(g["scores"]), atol=2e-5)
    m0, ms0 = lg.filter_matches(scores, 0.1)
    assert torch.equal(m0, _t(g["matches0"]).long())
    assert torch.allclose(ms0, _t(g["mscores0"]), atol=1e-6)
    assert torch.allclose(lg.matchability(SD, 5, d0), _t(g["matchability0"]), atol=1e-6)


def test_full_matcher_runs_prunes_and_is_consistent():
    k0, d0, s0, k1, d1, s1 = fixtures.lighterglue_inputs(300, 260, seed=3)
    trace = []
    m, sc = lg.lighterglue_forward(SD, k0, d0, s0, k1, d1, s1, min_conf=0.1, trace=trace)
    assert m.dtype == torch.int64 and m.shape[1] == 2 and sc.shape[0] == m.shape[0]
    assert (m[:, 0] < 300).all() and (m[:, 1] < 260).all() and len(set(m[:, 0].tolist())) == m.shape[0] and len(set(m[:, 1].tolist())) == m.shape[0]
    assert torch.equal(m[:, 0], m[:, 0].sort().values)                       # ascending in image 0, like torch.where
    sizes = [t[0].shape[0] for t in trace[:-1]]
    assert sizes[0] == 300 and sizes[-1] <= sizes[0]                        # width pruning only ever removes points
    m2, _ = lg.lighterglue_forward(SD, k0, d0, s0, k1, d1, s1, min_conf=0.1, prune=False)
    assert m2.shape[1] == 2                                                  # un-pruned variant (what a CUDA run below the threshold does)


def test_end_to_end_without_pruning_matches_the_hf_port():
    """The whole matcher (input projection, encoding, 6 layers, last-layer assignment, mutual filter) against the HuggingFace
    port's own LightGlueForKeypointMatching._match_image_pair with tied weights (tests/golden/lg_e2e.npz)."""
    g = np.load(os.path.join(fixtures.GOLDEN_DIR, "lg_e2e.npz"))
    t = lambda k: torch.from_numpy(g[k])
    m, sc = lg.lighterglue_forward(SD, t("k0"), t("desc0"), t("size0"), t("k1"), t("desc1"), t("size1"), min_conf=float(g["threshold"]), prune=False)
    m0 = g["matches0"]
    want = [(i, int(j)) for i, j in enumerate(m0) if j > -1]
    assert len(want) >= 5
    assert [tuple(r) for r in m.tolist()] == want
    np.testing.assert_allclose(sc.numpy(), g["mscores0"][m0 > -1], rtol=2e-5, atol=1e-7)
    # the port's matches1 is the same assignment seen from image 1
    m1 = g["matches1"]
    assert all(int(m1[j]) == i for i, j in want) and int((m1 > -1).sum()) == len(want)
