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
    for mc, tag in ((0.82, "082"), (0.55, "055")):     # the reference default (empty on this fixture) and a cut through the middle
        j0, j1 = O.match_mnn(outs["a"]["descriptors"], outs["b"]["descriptors"], mc)
        parity.compare_matches(ka[j0], kb[j1], ga[g[f"idx0_{tag}"]], gb[g[f"idx1_{tag}"]], orc)
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


def test_g5_baseline_config0_assets_pair():
    """BASELINE configs[0]: assets/ref.png <-> tgt.png (600x800 uint8 RGB, real-resize branch 600 -> 576) through the
    reference's match_xfeat / detectAndCompute / match / match_xfeat_star (make_golden.py G5)."""
    g = np.load(os.path.join(G, "g5_assets.npz"))
    sd = _sd()
    im0, im1 = g["img0"], g["img1"]
    assert im0.shape == (600, 800, 3) and im0.dtype == np.uint8
    outs, heats = [], []
    for t, im in (("0", im0), ("1", im1)):
        o, st = O.detect_and_compute(sd, O.parse_input(im), top_k=4096, keep=True)
        o = o[0]
        gk = g[f"kp{t}"]
        full_ref = {"keypoints": gk, "scores": g[f"sc{t}"], "descriptors": np.zeros((len(gk), 64), np.float32)}
        full = {"keypoints": o["keypoints"], "scores": o["scores"], "descriptors": torch.zeros(len(o["keypoints"]), 64)}
        rep = parity.compare_keypoints(full, full_ref, heat=st["heat"][0, 0], rw=st["rw"], rh=st["rh"])
        assert rep["n_test"] == 4096 and rep["exceptions"] == 0, rep
        key = {(float(x), float(y)): i for i, (x, y) in enumerate(o["keypoints"].numpy())}
        rows = [key[(float(x), float(y))] for x, y in gk[::8]]
        parity.assert_close(o["descriptors"][rows], g[f"desc{t}_every8"], 1e-5, "desc")
        outs.append(o)
    orc = {"kp0": outs[0]["keypoints"], "kp1": outs[1]["keypoints"], "d0": outs[0]["descriptors"], "d1": outs[1]["descriptors"]}
    k0, k1, _, _ = O.match_xfeat(sd, im0, im1, top_k=4096)
    rep = parity.compare_matches(k0, k1, g["m0"], g["m1"], orc)
    assert rep["n_test"] == len(g["m0"]) >= 500 and rep["differing_rows"] == 0, rep
    j0, j1 = O.match_mnn(outs[0]["descriptors"], outs[1]["descriptors"], 0.5)
    parity.compare_matches(outs[0]["keypoints"][j0], outs[1]["keypoints"][j1], g["kp0"][g["idx0_050"]], g["kp1"][g["idx1_050"]], orc)
    star = O.match_xfeat_star(sd, im0, im1, top_k=4096)[0]
    rep = parity.compare_star_rows(star, np.concatenate([g["star0"], g["star1"]], 1))
    assert rep["exceptions"] == 0, rep


def test_g6_match_xfeat_star_many_refined_rows():
    """match_xfeat_star on a noisy-copy pair: > 1000 refined rows per pair, compared row by row with the reference's."""
    g = np.load(os.path.join(G, "g6_star.npz"))
    sd = _sd()
    sa, sb = fixtures.star_pair(2, 320, 384, seed=41)
    d0 = O.detect_and_compute_dense(sd, sa, top_k=2048)
    d1 = O.detect_and_compute_dense(sd, sb, top_k=2048)
    parity.assert_close(d0["keypoints"], g["kp_a"], 1e-4, "dense kp a")
    parity.assert_close(d1["keypoints"], g["kp_b"], 1e-4, "dense kp b")
    parity.assert_close(d0["descriptors"][:, ::8], g["desc_a_every8"], 5e-5, "dense desc")
    bm = O.batch_match(d0["descriptors"], d1["descriptors"])
    res = O.match_xfeat_star(sd, sa, sb, top_k=2048)
    for b in range(2):
        pt = set(zip(map(tuple, d0["keypoints"][b][bm[b][0]].tolist()), map(tuple, d1["keypoints"][b][bm[b][1]].tolist())))
        pg = set(zip(map(tuple, g["kp_a"][b][g[f"bm{b}_idx0"]].tolist()), map(tuple, g["kp_b"][b][g[f"bm{b}_idx1"]].tolist())))
        assert pt == pg and len(pg) >= 1500          # as coordinates: the index order inside reliability ties is free
        rep = parity.compare_star_rows(res[b], g[f"star{b}"], {"sd": sd, "d0": d0, "d1": d1, "b": b})
        assert rep["n_ref"] >= 1000 and rep["exceptions"] == 0, rep
