"""Pin the oracle (oracle/xfeat_oracle.py) against vectors produced by the UNMODIFIED
reference (tests/golden/make_golden.py ran /root/reference/modules on CPU)."""
import os

import numpy as np
import torch

import fixtures
import parity
from oracle import xfeat_oracle as O

G = fixtures.GOLDEN_DIR
torch.set_num_threads(min(8, os.cpu_count() or 1))


def _sd():
    return fixtures.synthetic_state_dict(0)


def test_g1_network_outputs_and_sparse_small():
    g = np.load(os.path.join(G, "g1_small.npz"))
    sd = _sd()
    x = fixtures.texture_images(2, 96, 128, seed=11)
    feats, logits, rel = O.backbone(sd, x)
    parity.assert_close(feats, g["feats"], 2e-5, "feats")
    parity.assert_close(logits, g["logits"], 1e-4, "logits")
    parity.assert_close(rel, g["reliability"], 1e-5, "reliability")
    heat = O.kpts_heatmap(logits)
    parity.assert_close(heat, g["heat"], 1e-5, "heat")
    out, st = O.detect_and_compute(sd, x, top_k=256, keep=True)
    for b, o in enumerate(out):
        ref = {k: g[f"{k}{b}"] for k in ("keypoints", "scores", "descriptors")}
        rep = parity.compare_keypoints(o, ref, heat=st["heat"][b, 0])
        assert rep["common"] >= 250, rep


def test_g2_vga_pair_detect_and_match():
    g = np.load(os.path.join(G, "g2_vga_pair.npz"))
    sd = _sd()
    a, b = fixtures.shifted_pair(1, 480, 640, seed=7)
    outs = {}
    for tag, img in (("a", a), ("b", b)):
        o, st = O.detect_and_compute(sd, img, keep=True)
        o = o[0]
        outs[tag] = o
        kp = o["keypoints"].numpy()
        # golden descriptors are stored for every 8th row only: compare via coordinates
        gk = g[f"kp_{tag}"].astype(np.float32)
        ref = {"keypoints": gk[::8], "scores": g[f"sc_{tag}"][::8], "descriptors": g[f"desc_{tag}_every8"]}
        sub = {"keypoints": o["keypoints"][::8], "scores": o["scores"][::8], "descriptors": o["descriptors"][::8]}
        if np.array_equal(kp, gk):
            parity.assert_close(sub["descriptors"], ref["descriptors"], 1e-5, "desc")
            parity.assert_close(o["scores"], g[f"sc_{tag}"], 1e-5, "scores")
            parity.assert_close(o["descriptors"].double().sum(1), g[f"desc_{tag}_rowsum"], 1e-4, "rowsum")
        else:   # tie-aware path
            full_ref = {"keypoints": gk, "scores": g[f"sc_{tag}"],
                        "descriptors": np.zeros((len(gk), 64), np.float32)}
            full = dict(o)
            full["descriptors"] = torch.zeros(len(kp), 64)
            parity.compare_keypoints(full, full_ref, heat=st["heat"][0, 0])
    ka, kb = outs["a"]["keypoints"], outs["b"]["keypoints"]
    ga, gb = g["kp_a"].astype(np.float32), g["kp_b"].astype(np.float32)
    orc = {"kp0": ka, "kp1": kb, "d0": outs["a"]["descriptors"], "d1": outs["b"]["descriptors"]}
    i0, i1 = O.match_mnn(outs["a"]["descriptors"], outs["b"]["descriptors"], -1)
    assert torch.all(i0[1:] > i0[:-1])
    parity.compare_matches(ka[i0], kb[i1], ga[g["idx0"]], gb[g["idx1"]], orc)
    j0, j1 = O.match_mnn(outs["a"]["descriptors"], outs["b"]["descriptors"], 0.82)
    parity.compare_matches(ka[j0], kb[j1], ga[g["idx0_082"]], gb[g["idx1_082"]], orc)
    assert len(i0) > 1000 and len(j0) > 100


def test_g3_match_xfeat_numpy_uint8_resize_path():
    g = np.load(os.path.join(G, "g3_match_xfeat.npz"))
    sd = _sd()
    ta, tb = fixtures.shifted_pair(1, 200, 300, seed=21, shift=(5, 9))
    ia = (ta[0].permute(1, 2, 0).numpy() * 255).clip(0, 255).astype(np.uint8)
    ib = (tb[0].permute(1, 2, 0).numpy() * 255).clip(0, 255).astype(np.uint8)
    k0, k1, _, _ = O.match_xfeat(sd, ia, ib, top_k=1024)
    assert k0.shape == g["m0"].shape, (k0.shape, g["m0"].shape)
    parity.assert_close(k0, g["m0"], 1e-4, "mkpts0")
    parity.assert_close(k1, g["m1"], 1e-4, "mkpts1")


def test_g4_dense_and_refine():
    g = np.load(os.path.join(G, "g4_dense.npz"))
    sd = _sd()
    sa, sb = fixtures.shifted_pair(2, 160, 192, seed=31, shift=(8, 8))
    d0 = O.detect_and_compute_dense(sd, sa, top_k=512)
    parity.assert_close(d0["keypoints"], g["dense_kp"], 1e-4, "dense kp")
    parity.assert_close(d0["descriptors"], g["dense_desc"], 5e-5, "dense desc")
    parity.assert_close(d0["scales"], g["dense_scales"], 1e-6, "scales")
    d1 = O.detect_and_compute_dense(sd, sb, top_k=512)
    n = d0["keypoints"].shape[1]
    for b in range(2):
        forced = [(torch.arange(n), torch.from_numpy(g[f"forced_perm{b}"]))] * 2
        r = O.refine_matches(sd, d0, d1, forced, b)
        assert r.shape == g[f"refine{b}"].shape
        parity.assert_close(r, g[f"refine{b}"], 2e-4, "refine")
    res = O.match_xfeat_star(sd, sa, sb, top_k=512)
    for b in range(2):
        assert res[b].shape == g[f"star{b}"].shape
        parity.assert_close(res[b], g[f"star{b}"], 2e-4, "star")
