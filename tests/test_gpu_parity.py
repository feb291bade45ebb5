"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and against the
golden vectors produced by the unmodified reference (tests/golden/*.npz).

Bars (BASELINE.json north_star): key-point indices and match pairs bit-exact (tie-aware, see
tests/parity.py), descriptors / scores within 1e-4 fp32.  Run with `pytest -m gpu` on an MI355X.
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

import fixtures
import parity
from oracle import xfeat_oracle as O

pytestmark = pytest.mark.gpu
G = fixtures.GOLDEN_DIR
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL_ACT = 5e-5      # activations are O(1); fp32 summation-order noise is ~1e-6


@pytest.fixture(scope="module")
def sd():
    return fixtures.synthetic_state_dict(0)


@pytest.fixture(scope="module")
def xf(sd):
    from accelerated_features_amd import XFeat
    m = XFeat(weights=sd, top_k=4096, detection_threshold=0.05)
    assert m.dev.type == "cuda"
    return m


def _lib():
    from accelerated_features_amd import _lib as L
    return L


def test_library_is_the_hip_one(xf):
    L = _lib()
    lib = L.load()
    assert lib.xfh_version() >= 100
    assert os.path.basename(L.LIB_PATH) == "libxfeat_hip.so"
    assert xf.net.handle() is not None


# ----------------------------------------------------------------------------------------------
# every conv layer in isolation, fed with the oracle's own input activation
# ----------------------------------------------------------------------------------------------
_LAYER_IO = {   # layer name -> (oracle tensor feeding it, oracle tensor it produces)
    "block1.0": ("gray", "block1.0"), "block1.1": ("block1.0", "block1.1"), "block1.2": ("block1.1", "block1.2"),
    "block1.3": ("block1.2", "block1.3"), "block2.0": ("x1", "block2.0"), "block2.1": ("block2.0", "block2.1"),
    "block3.0": ("block2.1", "block3.0"), "block3.1": ("block3.0", "block3.1"), "block3.2": ("block3.1", "block3.2"),
    "block4.0": ("block3.2", "block4.0"), "block4.1": ("block4.0", "block4.1"), "block4.2": ("block4.1", "block4.2"),
    "block5.0": ("block4.2", "block5.0"), "block5.1": ("block5.0", "block5.1"), "block5.2": ("block5.1", "block5.2"),
    "block5.3": ("block5.2", "block5.3"), "block_fusion.0": ("pyramid", "block_fusion.0"),
    "block_fusion.1": ("block_fusion.0", "block_fusion.1"), "block_fusion.2": ("block_fusion.1", "feats"),
    "heatmap_head.0": ("feats", "heatmap_head.0"), "heatmap_head.1": ("heatmap_head.0", "heatmap_head.1"),
    "keypoint_head.0": ("unfold", "keypoint_head.0"), "keypoint_head.1": ("keypoint_head.0", "keypoint_head.1"),
    "keypoint_head.2": ("keypoint_head.1", "keypoint_head.2"), "keypoint_head.3": ("keypoint_head.2", "logits"),
}


@pytest.fixture(scope="module")
def acts(sd):
    out = {}
    for tag, (B, H, W, seed) in {"a": (2, 96, 128, 11), "b": (1, 160, 224, 12)}.items():
        x = fixtures.texture_images(B, H, W, seed=seed)
        out[tag] = (x, O.backbone(sd, x, keep=True)[3])
    return out


@pytest.mark.parametrize("variant", [0, 1])
def test_conv_layers_isolated(xf, acts, variant):
    from accelerated_features_amd.spec import CONV_INDEX, CONV_BY_NAME
    L = _lib()
    lib = L.load()
    h = xf.net.handle()
    report, bad = [], []
    for tag, (x, t) in acts.items():
        for name, (src, dst) in _LAYER_IO.items():
            c = CONV_BY_NAME[name]
            xin = t[src].cuda().contiguous()
            ref = t[dst]
            if name == "block1.3":
                ref = ref          # the isolated layer is conv+BN+ReLU only (skip add is tested in the backbone)
            B, _, Hin, Win = xin.shape
            out = torch.full(tuple(ref.shape), float("nan"), device="cuda")
            rc = lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(xin.data_ptr()), B, Hin, Win,
                                    C.c_void_p(out.data_ptr()), variant, None)
            if rc == -4 and variant == 0:     # heads run channels-last in the backbone: covered elsewhere
                continue
            assert rc == 0, (name, lib.xfh_last_error())
            torch.cuda.synchronize()
            d = float((out.cpu() - ref).abs().max())
            report.append((tag, name, d))
            if not d <= TOL_ACT:
                bad.append((tag, name, d))
    print(report)
    assert not bad, f"variant {variant}: layers off: {bad}"


def test_backbone_intermediate_free_outputs_small(xf, sd):
    g = np.load(os.path.join(G, "g1_small.npz"))
    x = fixtures.texture_images(2, 96, 128, seed=11)
    feats, logits, rel = xf.net(x.cuda())
    of, ol, orl = O.backbone(sd, x)
    heat = xf.get_kpts_heatmap(logits)
    errs = {
        "feats_vs_oracle": float((feats.cpu() - of).abs().max()),
        "logits_vs_oracle": float((logits.cpu() - ol).abs().max()),
        "rel_vs_oracle": float((rel.cpu() - orl).abs().max()),
        "feats_vs_golden": float(np.abs(feats.cpu().numpy() - g["feats"]).max()),
        "logits_vs_golden": float(np.abs(logits.cpu().numpy() - g["logits"]).max()),
        "rel_vs_golden": float(np.abs(rel.cpu().numpy() - g["reliability"]).max()),
        "heat_vs_golden": float(np.abs(heat.cpu().numpy() - g["heat"]).max()),
    }
    print(errs)
    assert feats.shape == of.shape and logits.shape == ol.shape and rel.shape == orl.shape
    assert errs["feats_vs_oracle"] <= 1e-4 and errs["feats_vs_golden"] <= 1e-4, errs
    assert errs["logits_vs_oracle"] <= 5e-4 and errs["logits_vs_golden"] <= 5e-4, errs   # logits are O(50)
    assert errs["rel_vs_oracle"] <= 1e-5 and errs["heat_vs_golden"] <= 1e-5, errs


FX_CONV64, FX_CONV24, FX_HEADS, FX_FINE = 1, 2, 8, 2048      # include/xfeat_hip.h: XFH_FX_*
FX_ALL = FX_CONV64 | FX_CONV24 | FX_HEADS | FX_FINE


@pytest.mark.parametrize("opts", [{"fx": 0, "block1": 5}, {"fx": 0}, {"block1": 5}, {"fx": FX_CONV64}, {"fx": FX_CONV24}, {"fx": FX_HEADS}, {"fx": FX_ALL & ~FX_CONV64}, {"fx": FX_ALL & ~FX_HEADS, "block1": 5}])
def test_backbone_fallback_kernels_same_results(opts, sd):
    """Per-handle switches (xfh_set_option) select, family by family, the fp32-range fallback instead of the fp16-pair default (the f32-MFMA convolutions and heads, block1 on the
    vector ALUs; {"fx": 0, "block1": 5} = the range fallback as a whole): the small golden backbone case on a model of its own with each setting."""
    from accelerated_features_amd import XFeat
    g = np.load(os.path.join(G, "g1_small.npz"))
    xf2 = XFeat(weights=sd, top_k=4096)
    for k, v in opts.items():
        xf2.set_option(k, v)
    x = fixtures.texture_images(2, 96, 128, seed=11)
    feats, logits, rel = xf2.net(x.cuda())
    heat = xf2.get_kpts_heatmap(logits)
    e = [float(np.abs(feats.cpu().numpy() - g["feats"]).max()), float(np.abs(logits.cpu().numpy() - g["logits"]).max()),
         float(np.abs(rel.cpu().numpy() - g["reliability"]).max()), float(np.abs(heat.cpu().numpy() - g["heat"]).max())]
    print("ERRS", opts, e)
    assert e[0] <= 1e-4 and e[1] <= 5e-4 and e[2] <= 1e-5 and e[3] <= 1e-5, (opts, e)
    with pytest.raises(Exception):
        xf2.set_option("no_such_option", 1)
    for key, value in (("wino", 1), ("bx", 21), ("heads_f32", 0), ("block1", 6), ("fx", 4), ("fx", 3979)):      # (deleted in round 6: the library refuses them too)
        with pytest.raises(Exception):
            xf2.set_option(key, value)
        assert _lib().load().xfh_set_option(xf2.net.handle(), key.encode(), value) != 0


@pytest.mark.parametrize("mode", [5, 7])
def test_block1_forms_alone(sd, acts, mode):
    """block1 + skip1 alone (xfh_debug_block1) in the vector form (5: the range fallback) and with block1.2 and block1.3 on the fp16 matrix cores (7: the default; fp16-pair
    arithmetic, csrc/block1_fx.hpp): against the oracle's x1 on its own normalised gray images, and form against form on shapes whose last tiles are partial
    (W / 4 = 88: five and a half 16-column tiles; H / 4 = 60: seven and a half 8-row tiles) -- with the position of the largest difference, for whoever has to debug it."""
    from accelerated_features_amd import XFeat
    lib = _lib().load()
    m = XFeat(weights=sd, top_k=512)
    m.set_option("block1", mode)
    ref = XFeat(weights=sd, top_k=512)
    ref.set_option("block1", 5)

    def run(model, gray):
        B, H, W = gray.shape
        coef = torch.tensor([[1.0, 0.0]] * B, device="cuda")
        x1 = torch.full((B, 24, H // 4, W // 4), float("nan"), device="cuda")
        assert lib.xfh_debug_block1(model.net.handle(), C.c_void_p(gray.data_ptr()), C.c_void_p(coef.data_ptr()), B, H, W, C.c_void_p(x1.data_ptr()), None) == 0, lib.xfh_last_error()
        torch.cuda.synchronize()
        return x1

    def where(d):
        i = int(d.argmax())
        return tuple(int(v) for v in np.unravel_index(i, tuple(d.shape)))
    for tag, (x, t) in acts.items():
        g = t["gray"].cuda().contiguous()[:, 0]
        got = run(m, g)
        d = (got.cpu() - t["x1"]).abs()
        assert bool(torch.isfinite(got).all()) and float(d.max()) <= TOL_ACT, (tag, mode, float(d.max()), where(d))
    assert m.net.take_status() == 0
    for B, H, W, seed in ((3, 224, 352, 5), (2, 240, 320, 6), (1, 32, 32, 7)):
        g = torch.randn(B, H, W, generator=torch.Generator().manual_seed(seed)).cuda()
        a, b = run(m, g), run(ref, g)
        d = (a - b).abs().cpu()
        assert bool(torch.isfinite(a).all()) and float(d.max()) <= 2e-5 * max(1.0, float(b.abs().max())), (mode, (B, H, W), float(d.max()), where(d))
    if mode == 7:      # the range guard of the pair: activations beyond 65504 in c2 / c3 are reported
        g = torch.randn(1, 64, 64, generator=torch.Generator().manual_seed(8)).cuda() * 1.0e7
        run(m, g)
        assert m.net.take_status() & 1


def test_default_backbone_equals_fallback_backbone_at_bench_shape(xf, sd):
    """At the benchmark shape (B=64 VGA) the default path runs every convolution from block1.2 on and both heads on the fp16 matrix cores in the fp16-pair arithmetic.  Same
    network outputs as with every one of them on the fp32-range kernels (a second model with fx = 0, block1 = 5: the range fallback)."""
    from accelerated_features_amd import XFeat
    xr = XFeat(weights=sd, top_k=4096)
    xr.set_option("fx", 0); xr.set_option("block1", 5)
    x = torch.cat([fixtures.texture_images(8, 480, 640, seed=s) for s in range(8)]).cuda()
    f_, l_, r_ = xr.net(x)
    ref = {"feats": f_[::8].cpu().numpy(), "logits": l_[::8].cpu().numpy(), "rel": r_.cpu().numpy()}
    del xr, f_, l_, r_
    xb = XFeat(weights=sd, top_k=4096)
    feats, logits, rel = xb.net(x)
    assert xb.net.take_status() == 0
    e = {"feats": float(np.abs(feats[::8].cpu().numpy() - ref["feats"]).max()), "logits": float(np.abs(logits[::8].cpu().numpy() - ref["logits"]).max()),
         "rel": float(np.abs(rel.cpu().numpy() - ref["rel"]).max())}
    print(e, "feats absmax", float(np.abs(ref["feats"]).max()), "logits absmax", float(np.abs(ref["logits"]).max()))
    assert e["feats"] <= 1e-4 and e["logits"] <= 5e-4 and e["rel"] <= 3e-5, e      # two fp32-accurate computations of a 20-layer network


def test_backbone_fused_heat_equals_helper(xf):
    x = fixtures.texture_images(2, 96, 128, seed=11).cuda()
    feats, logits, heat, rel = xf.net.backbone(x, want_logits=True, want_heat=True)
    h2 = xf.get_kpts_heatmap(logits.permute(0, 3, 1, 2))
    # same logits, two softmax evaluations (fused epilogue vs helper kernel): summation order differs
    assert float((heat - h2[:, 0]).abs().max()) <= 5e-7


def test_nms_helper_matches_oracle(xf, sd):
    x = fixtures.texture_images(2, 96, 128, seed=11)
    _, logits, _ = O.backbone(sd, x)
    heat = O.kpts_heatmap(logits)
    ref = O.pad_keypoints(O.nms(heat, 0.05, 5))
    got = xf.NMS(heat.cuda(), threshold=0.05, kernel_size=5).cpu()
    assert got.dtype == torch.int64 and got.shape == ref.shape
    assert torch.equal(got, ref)
    # plateau / strict threshold / row-major order
    hm = torch.zeros(1, 1, 32, 64)
    hm[0, 0, 3, 3] = 0.5; hm[0, 0, 3, 4] = 0.5; hm[0, 0, 10, 2] = 0.05; hm[0, 0, 12, 12] = 0.9
    hm[0, 0, 12, 14] = 0.8; hm[0, 0, 0, 63] = 0.3; hm[0, 0, 31, 0] = 0.2
    assert xf.NMS(hm.cuda()).cpu()[0].tolist() == O.nms(hm)[0].tolist() == [[63, 0], [3, 3], [4, 3], [12, 12], [0, 31]]


# ----------------------------------------------------------------------------------------------
# sparse path end to end
# ----------------------------------------------------------------------------------------------
def test_detect_small_vs_oracle_and_golden(xf, sd):
    g = np.load(os.path.join(G, "g1_small.npz"))
    x = fixtures.texture_images(2, 96, 128, seed=11)
    out = xf.detectAndCompute(x.cuda(), top_k=256)
    ref, st = O.detect_and_compute(sd, x, top_k=256, keep=True)
    for b in range(2):
        rep = parity.compare_keypoints(out[b], ref[b], heat=st["heat"][b, 0])
        gold = {k: g[f"{k}{b}"] for k in ("keypoints", "scores", "descriptors")}
        rep2 = parity.compare_keypoints(out[b], gold, heat=st["heat"][b, 0])
        print(b, rep, rep2)
        assert out[b]["keypoints"].dtype == torch.float32 and out[b]["descriptors"].shape[1] == 64


@pytest.fixture(scope="module")
def vga_pair(xf, sd):
    a, b = fixtures.shifted_pair(1, 480, 640, seed=7)
    res = {}
    for tag, img in (("a", a), ("b", b)):
        hip = xf.detectAndCompute(img.cuda())[0]
        orc, st = O.detect_and_compute(sd, img, keep=True)
        res[tag] = (hip, orc[0], st)
    return res


def test_detect_vga_top4096_vs_oracle_and_golden(vga_pair):
    g = np.load(os.path.join(G, "g2_vga_pair.npz"))
    for tag in ("a", "b"):
        hip, orc, st = vga_pair[tag]
        rep = parity.compare_keypoints(hip, orc, heat=st["heat"][0, 0])
        print(tag, "vs oracle", rep)
        assert rep["n_test"] == 4096
        gk = g[f"kp_{tag}"].astype(np.float32)
        gold = {"keypoints": gk, "scores": g[f"sc_{tag}"], "descriptors": np.zeros((len(gk), 64), np.float32)}
        t = {"keypoints": hip["keypoints"], "scores": hip["scores"], "descriptors": torch.zeros(len(hip["keypoints"]), 64)}
        rep = parity.compare_keypoints(t, gold, heat=st["heat"][0, 0])
        print(tag, "vs golden", rep)
        # golden descriptors (every 8th row of the reference's output), matched through coordinates
        key = {(float(x), float(y)): i for i, (x, y) in enumerate(hip["keypoints"].cpu().numpy())}
        rows = [(key[(float(x), float(y))], j) for j, (x, y) in enumerate(gk[::8]) if (float(x), float(y)) in key]
        it = torch.tensor([r[0] for r in rows]); jt = [r[1] for r in rows]
        d = float(np.abs(hip["descriptors"].cpu().numpy()[it.numpy()] - g[f"desc_{tag}_every8"][jt]).max())
        assert len(rows) >= 500 and d <= 1e-4, (len(rows), d)


def test_match_vga_vs_oracle_and_golden(xf, vga_pair):
    g = np.load(os.path.join(G, "g2_vga_pair.npz"))
    ha, oa, _ = vga_pair["a"]
    hb, ob, _ = vga_pair["b"]
    orc = {"kp0": oa["keypoints"], "kp1": ob["keypoints"], "d0": oa["descriptors"], "d1": ob["descriptors"]}
    ga, gb = g["kp_a"].astype(np.float32), g["kp_b"].astype(np.float32)
    for mc, k0, k1 in ((-1, "idx0", "idx1"), (0.82, "idx0_082", "idx1_082"), (0.55, "idx0_055", "idx1_055")):
        i0, i1 = xf.match(ha["descriptors"], hb["descriptors"], min_cossim=mc)
        assert i0.dtype == torch.int64 and i1.dtype == torch.int64
        assert torch.all(i0[1:] > i0[:-1])
        o0, o1 = O.match_mnn(oa["descriptors"], ob["descriptors"], mc)
        ka, kb = ha["keypoints"].cpu(), hb["keypoints"].cpu()
        rep = parity.compare_matches(ka[i0.cpu()], kb[i1.cpu()], oa["keypoints"][o0], ob["keypoints"][o1], orc)
        rep2 = parity.compare_matches(ka[i0.cpu()], kb[i1.cpu()], ga[g[k0]], gb[g[k1]], orc)
        print(mc, rep, rep2)
        assert rep["n_test"] == len(g[k0]) and (rep["n_test"] > 500 or mc == 0.82)


def test_match_same_inputs_as_oracle_exact_sizes(xf):
    """match() on identical descriptor inputs: ragged sizes, ties, min_cossim."""
    g = torch.Generator().manual_seed(3)
    for n1, n2 in ((300, 517), (1, 1), (33, 257), (256, 128), (1000, 31), (4096, 4096)):
        d1 = torch.nn.functional.normalize(torch.randn(n1, 64, generator=g), dim=-1)
        d2 = torch.nn.functional.normalize(torch.randn(n2, 64, generator=g), dim=-1)
        if n1 >= 300 and n2 >= 300:
            d2[7] = d1[5]; d2[8] = d1[5]        # exact duplicate columns -> arg-max tie: lowest index wins
            d1[100] = d1[101]                    # duplicate rows -> column tie: lowest row wins
        for mc in (-1, 0.3):
            o0, o1 = O.match_mnn(d1, d2, mc)
            i0, i1 = xf.match(d1.cuda(), d2.cuda(), min_cossim=mc)
            s = (d1.double() @ d2.double().t())
            same = torch.equal(i0.cpu(), o0) and torch.equal(i1.cpu(), o1)
            if not same:   # only near-ties may differ
                a = {int(x): int(y) for x, y in zip(i0.cpu(), i1.cpu())}
                b = {int(x): int(y) for x, y in zip(o0, o1)}
                diff = [k for k in set(a) | set(b) if a.get(k) != b.get(k)]
                for k in diff:
                    top = torch.topk(s[k], min(2, n2))[0]
                    col = s[:, a.get(k, b.get(k))]
                    ctop = torch.topk(col, min(2, n1))[0]
                    gap = min(float(top[0] - top[-1]), float(ctop[0] - ctop[-1]))
                    assert gap <= 2e-6 or abs(float(top[0]) - mc) <= 2e-6, (n1, n2, mc, k, a.get(k), b.get(k), gap)
                assert len(diff) <= max(2, n1 // 200), (n1, n2, len(diff))
    e0, e1 = xf.match(torch.zeros(0, 64).cuda(), torch.zeros(5, 64).cuda())
    assert len(e0) == 0 and len(e1) == 0


def test_match_xfeat_resize_path_vs_golden(xf, sd):
    """numpy uint8 HWC 200x300 -> real-resize branch (192x288): match_xfeat rows against the reference's golden rows and
    the oracle, pair for pair; a differing pair must be a near-tie of the oracle's similarity matrix (tests/parity.py)."""
    g = np.load(os.path.join(G, "g3_match_xfeat.npz"))
    ta, tb = fixtures.shifted_pair(1, 200, 300, seed=21, shift=(5, 9))
    ia = (ta[0].permute(1, 2, 0).numpy() * 255).clip(0, 255).astype(np.uint8)
    ib = (tb[0].permute(1, 2, 0).numpy() * 255).clip(0, 255).astype(np.uint8)
    m0, m1 = xf.match_xfeat(ia, ib, top_k=1024)
    assert isinstance(m0, np.ndarray) and m0.dtype == np.float32 and m0.shape[1] == 2
    oa = O.detect_and_compute(sd, O.parse_input(ia).float(), 1024)[0]
    ob = O.detect_and_compute(sd, O.parse_input(ib).float(), 1024)[0]
    orc = {"kp0": oa["keypoints"], "kp1": ob["keypoints"], "d0": oa["descriptors"], "d1": ob["descriptors"]}
    rep = parity.compare_matches(m0, m1, g["m0"], g["m1"], orc)
    print("g3", rep)
    assert rep["n_ref"] == len(g["m0"]) and rep["n_test"] >= 300


def test_preprocess_resize_matches_oracle(xf):
    x = fixtures.texture_images(2, 200, 300, seed=5)
    y, rh, rw = xf.preprocess_tensor(x)
    ry, rrh, rrw = O.preprocess(x)
    assert (rh, rw) == (rrh, rrw) and tuple(y.shape) == tuple(ry.shape)
    assert float((y.cpu() - ry).abs().max()) <= 1e-6
    z, rh, rw = xf.preprocess_tensor(x[:, :, :192, :288])
    assert torch.equal(z.cpu(), x[:, :, :192, :288]) and rh == 1.0 and rw == 1.0
    g = (torch.rand(37, 65) * 255).to(torch.uint8).numpy()          # (H,W) numpy uint8, NOT divided by 255
    y, _, _ = xf.preprocess_tensor(g)
    ry, _, _ = O.preprocess(torch.tensor(g[..., None]).permute(2, 0, 1)[None])
    assert float((y.cpu() - ry).abs().max()) <= 1e-4
    with pytest.raises(RuntimeError):
        xf.preprocess_tensor(torch.zeros(3, 64, 64))


# ----------------------------------------------------------------------------------------------
# edge cases of the sparse path
# ----------------------------------------------------------------------------------------------
def test_detect_edge_cases(xf, sd):
    # no key-points at all (threshold above every heat value) -> empty, well-typed results
    x = fixtures.texture_images(2, 64, 96, seed=2)
    out = xf.detectAndCompute(x.cuda(), top_k=128, detection_threshold=2.0)
    for o in out:
        assert o["keypoints"].shape == (0, 2) and o["scores"].shape == (0,) and o["descriptors"].shape == (0, 64)
    # ragged batch: different images, fewer candidates than top_k
    x = fixtures.texture_images(3, 64, 96, seed=4)
    x[1] *= 0.0                                    # constant image: instance-norm -> all zeros
    out = xf.detectAndCompute(x.cuda(), top_k=4096, detection_threshold=0.05)
    ref, st = O.detect_and_compute(sd, x, top_k=4096, detection_threshold=0.05, keep=True)
    for b in range(3):
        rep = parity.compare_keypoints(out[b], ref[b], heat=st["heat"][b, 0])
        print("ragged", b, rep)
    # capacity overflow path: force a tiny NMS capacity and check the exact re-run
    kp, sc, de, nv, nc, cap, hw = xf._detect_device(x.cuda(), 4096, 0.05, cap=16)
    assert int(nc.max()) > 16
    # grayscale (C=1) and C=3 with identical channels agree
    g1 = fixtures.texture_images(1, 64, 96, seed=9, channels=1)
    o1 = xf.detectAndCompute(g1.cuda(), top_k=64)[0]
    o3 = xf.detectAndCompute(g1.repeat(1, 3, 1, 1).cuda(), top_k=64)[0]
    r1 = O.detect_and_compute(sd, g1, top_k=64)[0]
    parity.compare_keypoints(o1, r1, heat=O.detect_and_compute(sd, g1, top_k=64, keep=True)[1]["heat"][0, 0])
    assert len(o1["keypoints"]) == len(o3["keypoints"])
    # last row / last column key-points get score 0 and are dropped: no returned point on them
    for o in out:
        if len(o["keypoints"]):
            assert float(o["keypoints"][:, 0].max()) < 95 and float(o["keypoints"][:, 1].max()) < 63
            assert float(o["scores"].min()) > 0


def test_batch_sizes_cover_both_workgroup_mappings(xf):
    """Batches that are / are not multiples of 8 take different workgroup->work mappings (XCD-aware vs
    plain); per-image results must be bit-identical in all of them, for the sparse path and the matcher."""
    u = fixtures.texture_images(5, 96, 160, seed=23).cuda()
    ref = [xf.detectAndCompute(u[i:i + 1], top_k=300)[0] for i in range(5)]
    for B in (2, 8, 9, 16, 24):
        idx = [i % 5 for i in range(B)]
        out = xf.detectAndCompute(u[idx], top_k=300)
        for j, i in enumerate(idx):
            for k in ("keypoints", "scores", "descriptors"):
                assert torch.equal(out[j][k], ref[i][k]), (B, j, k)
    d = torch.nn.functional.normalize(torch.randn(24, 700, 64, device="cuda"), dim=-1)
    single = [xf.match(d[p], d[(p + 1) % 24], min_cossim=-1) for p in range(24)]
    for P in (3, 8, 16, 24):
        bm = xf.batch_match(d[:P], torch.roll(d, -1, 0)[:P])
        for p in range(P):
            assert torch.equal(bm[p][0], single[p][0]) and torch.equal(bm[p][1], single[p][1]), (P, p)


def test_match_many_is_match_pair_by_pair(xf):
    """XFeat.match_many (the list form of match: one launch sequence, one read-back) returns, pair by pair, exactly what XFeat.match returns -- on the descriptors
    detectAndCompute hands out (matched in place: views of one padded tensor at a constant stride, fp16 copies reused), on the same lists in another order
    (irregular stride: the padded path), on free-standing tensors of different lengths, with an empty set, and with and without the similarity test."""
    x = torch.cat([fixtures.texture_images(4, 96, 160, seed=41), fixtures.texture_images(4, 96, 160, seed=41).roll((3, 5), (2, 3))]).cuda()
    res = xf.detectAndCompute(x, top_k=400)
    f1, f2 = [r["descriptors"] for r in res[0::2]], [r["descriptors"] for r in res[1::2]]
    assert xf._strided_layout(f1, f2) is not None and xf._strided_layout(f1, f2)[4] is None        # in place; the fp16 filter copies are made from the rows as they are NOW
    # ... so descriptors modified in place between detectAndCompute and match_many are matched AS MODIFIED (round 5 reused fp16 copies of the old rows)
    keep0 = f1[0].clone()
    with torch.inference_mode():              # (detectAndCompute hands out inference tensors, as the reference does)
        f1[0].copy_(torch.nn.functional.normalize(torch.roll(f1[0], 7, dims=1) + 0.1, dim=-1))
    want_mod = xf.match(f1[0], f2[0], min_cossim=-1)
    got_mod = xf.match_many(f1, f2, min_cossim=-1)[0]
    assert torch.equal(got_mod[0], want_mod[0]) and torch.equal(got_mod[1], want_mod[1])
    with torch.inference_mode():
        f1[0].copy_(keep0)
    for mc in (-1, 0.82):
        want = [xf.match(a, b, min_cossim=mc) for a, b in zip(f1, f2)]
        for got in (xf.match_many(f1, f2, min_cossim=mc),                                              # in place
                    xf.match_many([t.clone() for t in f1], [t.clone() for t in f2], min_cossim=mc)):      # padded copies
            assert len(got) == len(want)
            for (g0, g1), (w0, w1) in zip(got, want):
                assert g0.dtype == torch.int64 and torch.equal(g0, w0) and torch.equal(g1, w1)
    order = [3, 0, 2, 1]                                                                                # no constant stride: the padded path
    assert xf._strided_layout([f1[i] for i in order], [f2[i] for i in order]) is None
    got = xf.match_many([f1[i] for i in order], [f2[i] for i in order], min_cossim=-1)
    for i, (g0, g1) in zip(order, got):
        w0, w1 = xf.match(f1[i], f2[i], min_cossim=-1)
        assert torch.equal(g0, w0) and torch.equal(g1, w1)
    # the same frames against a shifted partner list (frame 2p against frame 2p + 3: another constant stride and offset)
    g1l, g2l = [r["descriptors"] for r in res[0:5:2]], [r["descriptors"] for r in res[3:8:2]]
    for (g0, g1), (a, b) in zip(xf.match_many(g1l, g2l, min_cossim=-1), zip(g1l, g2l)):
        w0, w1 = xf.match(a, b, min_cossim=-1)
        assert torch.equal(g0, w0) and torch.equal(g1, w1)
    d = [torch.nn.functional.normalize(torch.randn(n, 64, device="cuda"), dim=-1) for n in (700, 33, 0, 257)]
    e = [torch.nn.functional.normalize(torch.randn(n, 64, device="cuda"), dim=-1) for n in (512, 700, 40, 1)]
    got = xf.match_many(d, e, min_cossim=-1)
    for (g0, g1), (a, b) in zip(got, zip(d, e)):
        w0, w1 = xf.match(a, b, min_cossim=-1)
        assert torch.equal(g0, w0) and torch.equal(g1, w1), (len(a), len(b))
    assert xf.match_many([], []) == []


def test_batch_composition_and_determinism(xf):
    """An image's result does not depend on its batch neighbours, and reruns are bit-identical."""
    x = fixtures.texture_images(5, 96, 160, seed=17).cuda()
    full = xf.detectAndCompute(x, top_k=512)
    again = xf.detectAndCompute(x, top_k=512)
    for b in range(5):
        single = xf.detectAndCompute(x[b:b + 1], top_k=512)[0]
        for k in ("keypoints", "scores", "descriptors"):
            assert torch.equal(full[b][k], again[b][k]), k
            assert torch.equal(full[b][k], single[k]), (b, k)


# ----------------------------------------------------------------------------------------------
# semi-dense path
# ----------------------------------------------------------------------------------------------
def test_dense_extract_refine_star_vs_oracle_and_golden(xf, sd):
    g = np.load(os.path.join(G, "g4_dense.npz"))
    sa, sb = fixtures.shifted_pair(2, 160, 192, seed=31, shift=(8, 8))
    d0 = xf.detectAndComputeDense(sa.cuda(), top_k=512)
    o0 = O.detect_and_compute_dense(sd, sa, top_k=512)
    assert d0["keypoints"].shape == o0["keypoints"].shape == g["dense_kp"].shape
    for b in range(2):
        rep = parity.compare_dense(d0, {"keypoints": g["dense_kp"], "descriptors": g["dense_desc"], "scales": g["dense_scales"]}, b)
        print("dense vs golden", b, rep)
        rep = parity.compare_dense(d0, o0, b)
        assert rep["only_test"] == 0, rep
    # refine with the golden's forced index lists on the ORACLE's dense features (same inputs both sides)
    o1 = O.detect_and_compute_dense(sd, sb, top_k=512)
    n = o0["keypoints"].shape[1]
    dev0 = {k: v.cuda() for k, v in o0.items()}
    dev1 = {k: v.cuda() for k, v in o1.items()}
    for b in range(2):
        forced = [(torch.arange(n), torch.from_numpy(g[f"forced_perm{b}"]))] * 2
        r_or = O.refine_matches(sd, o0, o1, forced, b)
        r = xf.refine_matches(dev0, dev1, [(a.cuda(), c.cuda()) for a, c in forced], b).cpu()
        assert r.shape == r_or.shape, (r.shape, r_or.shape)
        parity.assert_close(r, r_or, 2e-4, "refine vs oracle")
        if r.shape == g[f"refine{b}"].shape:
            parity.assert_close(r, g[f"refine{b}"], 5e-4, "refine vs golden")
    # fine_matcher module call
    v = torch.cat([o0["descriptors"][0], o1["descriptors"][0]], -1)
    parity.assert_close(xf.net.fine_matcher(v.cuda()).cpu(), O.fine_matcher(sd, v), 2e-4, "fine_matcher")
    # batch_match on identical inputs
    bm = xf.batch_match(dev0["descriptors"], dev1["descriptors"])
    bo = O.batch_match(o0["descriptors"], o1["descriptors"])
    for b in range(2):
        assert torch.equal(bm[b][0].cpu(), bo[b][0]) and torch.equal(bm[b][1].cpu(), bo[b][1]), b
    # whole match_xfeat_star: every row against the reference's golden rows and the oracle's
    res = xf.match_xfeat_star(sa.cuda(), sb.cuda(), top_k=512)
    ref = O.match_xfeat_star(sd, sa, sb, top_k=512)
    d1h = xf.detectAndComputeDense(sb.cuda(), top_k=512)
    assert isinstance(res, list) and len(res) == 2
    for b in range(2):
        ctx = {"sd": sd, "d0": o0, "d1": o1, "b": b}
        dense = (d0["keypoints"][b], d1h["keypoints"][b])
        rep = parity.compare_star_rows(res[b], ref[b], ctx, test_dense=dense)
        rep2 = parity.compare_star_rows(res[b], g[f"star{b}"], ctx, test_dense=dense)
        print("star", b, rep, rep2)
        assert res[b].shape[1] == 4 and rep["n_ref"] == len(g[f"star{b}"]) >= 50


# ----------------------------------------------------------------------------------------------
# full BASELINE size: properties that need no oracle run (SURVEY 8c / task section 3)
# ----------------------------------------------------------------------------------------------
def test_full_size_vga_batch64_properties(xf):
    B = 64
    base = fixtures.texture_images(8, 480, 640, seed=101)
    x = torch.cat([base, torch.roll(base, (8, 16), (2, 3)), base.flip(3), base.flip(2),
                   base * 0.5 + 0.1, torch.roll(base, (-7, 3), (2, 3)), base.flip(2).flip(3), base]).cuda()
    assert x.shape[0] == B
    kp, sc, de, nv, nc, cap, hw = xf._detect_device(x, 4096, 0.05)
    assert int(nc.max()) <= cap
    nvl = nv.cpu().tolist()
    assert min(nvl) > 1000
    # sortedness, validity prefix, unit norm, bounds
    for b in range(0, B, 7):
        n = nvl[b]
        s = sc[b, :n]
        assert torch.all(s[1:] <= s[:-1]) and float(s.min()) > 0
        assert torch.allclose(de[b, :n].norm(dim=-1), torch.ones(n, device="cuda"), atol=1e-5)
        assert float(kp[b, :n, 0].max()) <= 638 and float(kp[b, :n, 1].max()) <= 478
    # the last 8 images repeat the first 8: identical results (batch-position independence)
    for b in range(8):
        assert nvl[b] == nvl[56 + b]
        assert torch.equal(kp[b], kp[56 + b]) and torch.equal(de[b], de[56 + b])
    # scaled+shifted intensities are removed by InstanceNorm up to rounding: same key-point count class
    assert abs(nvl[32] - nvl[0]) <= 64
    # pair matching on device counts; verify mutual-NN property against a torch matmul on the device
    i0, i1, nm = xf.match_pairs_device(de, nv, -1)
    nml = nm.cpu().tolist()
    for p in (0, 13, 31):
        a, b_ = de[2 * p, :nvl[2 * p]], de[2 * p + 1, :nvl[2 * p + 1]]
        s = a @ b_.t()
        r12, r21 = s.argmax(1), s.argmax(0)
        mutual = (r21[r12] == torch.arange(len(a), device="cuda")).nonzero()[:, 0]
        got0, got1 = i0[p, :nml[p]], i1[p, :nml[p]]
        assert torch.all(got0[1:] > got0[:-1])
        # same pair list up to arg-max near-ties
        sa_ = set(zip(got0.tolist(), got1.tolist()))
        sb_ = set(zip(mutual.tolist(), r12[mutual].tolist()))
        assert len(sa_ ^ sb_) <= max(4, len(sb_) // 200), (p, len(sa_), len(sb_), len(sa_ ^ sb_))
    # image 56 is a copy of image 0: the mutual matches are exactly the identity
    j0, j1 = xf.match(de[0, :nvl[0]], de[56, :nvl[56]], min_cossim=-1)
    assert len(j0) >= nvl[0] - 4 and torch.equal(j0, j1), (len(j0), nvl[0])


def test_repeated_launches_every_mfma_layer_no_intermittent_errors(xf):
    """Race screen: every MFMA conv layer, 25 launches each at two scales, every launch checked against the
    generic direct kernel on the device (an LDS-DMA / barrier race shows up as rare wrong tiles)."""
    from accelerated_features_amd.spec import CONVS, CONV_INDEX
    lib = _lib().load()
    h = xf.net.handle()
    div = {"block2.0": 4, "block2.1": 4, "block3.0": 4, "block3.1": 8, "block3.2": 8, "block4.0": 8, "block4.1": 16, "block4.2": 16,
           "block5.0": 16, "block5.1": 32, "block5.2": 32, "block5.3": 32, "block_fusion.0": 8, "block_fusion.1": 8, "block_fusion.2": 8}
    g = torch.Generator(device="cuda").manual_seed(1)
    for (B, H, W) in ((64, 480, 640), (8, 1312, 1312)):
        for name, d in div.items():
            c = next(c for c in CONVS if c.name == name)
            hin, win = H // d, W // d
            hout, wout = (hin - 1) // c.stride + 1, (win - 1) // c.stride + 1
            x = torch.randn(B, c.cin, hin, win, device="cuda", generator=g)
            ref = torch.empty(B, c.cout, hout, wout, device="cuda")
            assert lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(x.data_ptr()), B, hin, win, C.c_void_p(ref.data_ptr()), 1, None) == 0
            for rep in range(25):
                y = torch.full_like(ref, float("nan"))
                assert lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(x.data_ptr()), B, hin, win, C.c_void_p(y.data_ptr()), 0, None) == 0
                err = float((y - ref).abs().nan_to_num(1e9).max())
                assert err <= 2e-4, (name, (B, hin, win), rep, err)


def test_repeated_backbone_and_sparse_path_bit_identical(xf):
    """30 repetitions of the full sparse step at the BASELINE batch: results must be bit-identical every time."""
    x = fixtures.texture_images(8, 480, 640, seed=77)
    x = torch.cat([x] * 8).cuda()
    kp0, sc0, de0, nv0, nc0, cap, hw = xf._detect_device(x, 4096, 0.05)
    i00, i10, nm0 = xf.match_pairs_device(de0, nv0, -1)
    n0 = nm0.cpu().tolist()
    for rep in range(30):
        kp, sc, de, nv, nc, cap, hw = xf._detect_device(x, 4096, 0.05)
        i0, i1, nm = xf.match_pairs_device(de, nv, -1)
        assert torch.equal(kp, kp0) and torch.equal(sc, sc0) and torch.equal(de, de0) and torch.equal(nv, nv0), rep
        assert nm.cpu().tolist() == n0, rep
        for p in (0, 7, 31):
            assert torch.equal(i0[p, :n0[p]], i00[p, :n0[p]]) and torch.equal(i1[p, :n0[p]], i10[p, :n0[p]]), (rep, p)


V_DEFAULT, V_GENERIC, V_FX, V_F32, V_FX_PAIR, V_FX_PAIR_NHWC = 0, 1, 2, 3, 4, 5      # include/xfeat_hip.h: XFH_CONV_VARIANT_*


def test_f32_mfma_fallback_kernels_match_generic_kernel(xf):
    """The f32-MFMA kernel of every convolution layer (xfh_conv_layer variant F32: what the range fallback runs) against the generic direct kernel, on odd / small /
    non-multiple-of-8 shapes (scalar-store path, clipped regions, B not a multiple of 8)."""
    from accelerated_features_amd.spec import CONVS, CONV_INDEX
    lib = _lib().load()
    h = xf.net.handle()
    g = torch.Generator(device="cuda").manual_seed(3)
    n_checked = 0
    for name in ("block2.0", "block3.0", "block3.1", "block4.0", "block5.0", "block5.1", "block5.3"):
        c = next(c for c in CONVS if c.name == name)
        for (B, hh, ww) in ((3, 41, 41), (2, 60, 80), (9, 30, 40), (8, 15, 20), (1, 6, 10)):
            x = torch.randn(B, c.cin, hh, ww, device="cuda", generator=g)
            ho, wo = (hh - 1) // c.stride + 1, (ww - 1) // c.stride + 1
            ref = torch.empty(B, c.cout, ho, wo, device="cuda")
            assert lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(x.data_ptr()), B, hh, ww, C.c_void_p(ref.data_ptr()), V_GENERIC, None) == 0
            for variant in (V_DEFAULT, V_F32):
                y = torch.full_like(ref, float("nan"))
                rc = lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(x.data_ptr()), B, hh, ww, C.c_void_p(y.data_ptr()), variant, None)
                assert rc == 0, (name, variant, lib.xfh_last_error())
                err = float((y - ref).abs().nan_to_num(1e9).max())
                assert err <= 2e-4, (name, (B, hh, ww), variant, err)
                n_checked += 1
    assert n_checked == 7 * 5 * 2
    y = torch.empty(1, 24, 8, 8, device="cuda")
    assert lib.xfh_conv_layer(h, CONV_INDEX["block2.0"], C.c_void_p(y.data_ptr()), 1, 8, 8, C.c_void_p(y.data_ptr()), 10, None) != 0      # (round 5's variant numbers are gone)


def test_fp16_pair_conv_is_fp32_accurate(xf, sd):
    """The 3x3 layers from block2.0 on run on the fp16 matrix cores in the fp16-pair arithmetic (xfh_conv_layer variant FX: conv_bx_kernel / conv_bxs2_kernel, conv_rs64_kernel in
    its 64- and 128-channel forms, conv_bx64s2x_kernel): against an fp64 convolution of the same folded weights the error must stay at the level of the fp32 direct kernel
    (variant GENERIC) -- and of the f32-MFMA fallback kernel (variant F32) --, on odd / small / clipped shapes, large magnitudes, and batches that are not a multiple of 8."""
    from accelerated_features_amd.spec import CONVS, CONV_INDEX, BN_EPS
    lib = _lib().load()
    h = xf.net.handle()
    g = torch.Generator(device="cuda").manual_seed(5)
    n = 0
    for name in ("block2.0", "block2.1", "block3.0", "block_fusion.0", "block4.1", "block4.0", "block5.0", "block5.1"):
        c = next(c for c in CONVS if c.name == name)
        w = sd[f"{name}.layer.0.weight"].double().cuda()
        rm, rv = sd[f"{name}.layer.1.running_mean"].double().cuda(), sd[f"{name}.layer.1.running_var"].double().cuda()
        sc = 1.0 / torch.sqrt(rv + BN_EPS)
        wf, bf = (w * sc[:, None, None, None]).float().double(), (-rm * sc).float().double()      # the library folds in fp64, rounds to fp32
        for (B, hh, ww) in ((2, 24, 32), (3, 41, 41), (1, 6, 10), (9, 30, 40), (8, 120, 160), (2, 7, 70), (1, 1, 1), (16, 9, 33)):
            for scale in (1.0, 1e-3, 300.0):
                x = torch.randn(B, c.cin, hh, ww, device="cuda", generator=g) * scale
                truth = torch.relu(torch.nn.functional.conv2d(x.double(), wf, bf, stride=c.stride, padding=1))
                ref = float(truth.abs().max())
                err = {}
                for variant in (V_GENERIC, V_FX, V_F32):
                    y = torch.full(tuple(truth.shape), float("nan"), device="cuda")
                    rc = lib.xfh_conv_layer(h, CONV_INDEX[name], C.c_void_p(x.data_ptr()), B, hh, ww, C.c_void_p(y.data_ptr()), variant, None)
                    if rc != 0 and variant == V_FX and c.cin == 128 and ww > 61:
                        continue                      # (conv_rs64_kernel's 128-channel form takes maps up to 61 columns; the backbone then runs the f32 kernel)
                    assert rc == 0, (name, variant, lib.xfh_last_error())
                    err[variant] = float((y.double() - truth).abs().nan_to_num(1e9).max()) / ref
                if V_FX in err:
                    assert err[V_FX] <= max(2.0 * err[V_GENERIC], 1e-6), (name, (B, hh, ww), scale, err)
                assert err[V_F32] <= max(4.0 * err[V_GENERIC], 5e-6), (name, (B, hh, ww), scale, err)      # (v_mfma_f32_32x32x2_f32 sums K in another order than the fma chain: a few ulps more)
                n += 1
    assert n == 8 * 8 * 3
    assert xf.net.take_status() == 0                      # |x| stayed far below the fp16 range: no range report


def test_weights_beyond_the_pair_range_run_on_the_f32_kernels(sd):
    """A layer with a weight of magnitude >= 31 has no fp16-pair image (2^11 w would not be an fp16 number: xfh_create leaves it out) and must run on its fp32-range kernel
    whatever the options say (the heads used to drop to a retired kernel there).  One oversized weight in each family -- a head layer, a 64-channel 3x3, a
    24-channel 3x3, block1.3 -- on a model with the DEFAULT options: finite outputs equal to the oracle's within the usual tolerances (an fp16 image of such a weight holds
    inf: the pair kernels would deliver NaN), no range flag; and xfh_conv_layer's FX variant refuses those layers instead of falling back silently."""
    from accelerated_features_amd import XFeat
    from accelerated_features_amd.spec import CONV_INDEX
    lib = _lib().load()
    x = fixtures.texture_images(2, 96, 128, seed=11)
    for name in ("keypoint_head.1", "heatmap_head.0", "block3.1", "block4.0", "block5.1", "block2.1", "block1.3"):
        sd2 = {k: v.clone() for k, v in sd.items()}
        w = sd2[f"{name}.layer.0.weight"]
        sc = float(torch.sqrt(sd2[f"{name}.layer.1.running_var"][3] + 1e-5))
        w[3, 5, w.shape[2] // 2, w.shape[3] // 2] = 40.0 * sc          # the FOLDED weight (BatchNorm scale applied) is 40
        m = XFeat(weights=sd2, top_k=256)
        feats, logits, rel = m.net(x.cuda())
        of, ol, orl = O.backbone(sd2, x)
        e = (float((feats.cpu() - of).abs().max()) / max(1.0, float(of.abs().max())), float((logits.cpu() - ol).abs().max()) / max(1.0, float(ol.abs().max())), float((rel.cpu() - orl).abs().max()))
        print(name, e)
        assert bool(torch.isfinite(feats).all() and torch.isfinite(logits).all() and torch.isfinite(rel).all()), name
        assert e[0] <= 1e-4 and e[1] <= 1e-4 and e[2] <= 3e-5, (name, e)
        assert m.net.take_status() == 0, name
        if name.startswith("block") and name != "block1.3":
            c = m.net.state_dict()[f"{name}.layer.0.weight"].shape
            xin = torch.randn(1, c[1], 16, 24, device="cuda")
            y = torch.empty(1, c[0], 16, 24, device="cuda")
            assert lib.xfh_conv_layer(m.net.handle(), CONV_INDEX[name], C.c_void_p(xin.data_ptr()), 1, 16, 24, C.c_void_p(y.data_ptr()), V_FX, None) == -4, name      # XFH_ERR_UNSUPPORTED
            assert lib.xfh_conv_layer(m.net.handle(), CONV_INDEX[name], C.c_void_p(xin.data_ptr()), 1, 16, 24, C.c_void_p(y.data_ptr()), V_DEFAULT, None) == 0, name


def test_fp16_pair_arithmetic_reports_its_range_and_the_model_falls_back(sd):
    """The fp16-pair convolutions (option fx, the default) hold activations below 65504 only.  An input that drives a layer beyond that sets bit 0 of the
    handle's status word (xfh_set_status_buffer); detectAndCompute then repeats the call on the fp32-range kernels (fx = 0, block1 = 5) -- same results as a model
    that ran those from the start -- and the model stays there.  Small subnormal-range activations are exact in both."""
    import warnings
    from accelerated_features_amd import XFeat
    from accelerated_features_amd.xfeat import DEFAULT_FX, DEFAULT_BLOCK1
    from accelerated_features_amd.spec import CONV_INDEX
    lib = _lib().load()
    a, b = XFeat(weights=sd, top_k=512), XFeat(weights=sd, top_k=512)
    v = C.c_int(-1)
    for key, mirror in ((b"fx", DEFAULT_FX), (b"block1", DEFAULT_BLOCK1)):      # the Python mirrors of the library defaults
        assert lib.xfh_get_option(a.net.handle(), key, C.byref(v)) == 0 and v.value == mirror, (key, v.value, mirror)
    b.set_option("fx", 0); b.set_option("block1", 5)      # the fp32-range kernels the fallback lands on
    # (a) a single layer: 1e6-sized activations -> flag, and inf / nan in the fx output; the f32-MFMA kernel is fine
    x = torch.relu(torch.randn(2, 64, 24, 32, device="cuda")) * 3.0e5
    y = torch.empty(2, 64, 24, 32, device="cuda")
    assert lib.xfh_conv_layer(a.net.handle(), CONV_INDEX["block_fusion.0"], C.c_void_p(x.data_ptr()), 2, 24, 32, C.c_void_p(y.data_ptr()), V_FX, None) == 0
    assert a.net.take_status() & 1 and a.net.take_status() == 0
    assert lib.xfh_conv_layer(a.net.handle(), CONV_INDEX["block_fusion.0"], C.c_void_p(x.data_ptr()), 2, 24, 32, C.c_void_p(y.data_ptr()), V_F32, None) == 0
    assert a.net.take_status() == 0 and bool(torch.isfinite(y).all())
    assert lib.xfh_conv_layer(a.net.handle(), CONV_INDEX["block2.0"], C.c_void_p((x[:, :24] * 1.0).contiguous().data_ptr()), 2, 24, 32, C.c_void_p(y.data_ptr()), V_FX, None) == 0
    assert a.net.take_status() & 1
    # (b) end to end: weights whose block1 output is huge (skip1 bias) -> the fx layers overflow, the call is repeated on the fp32-range kernels, results = model b's
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["skip1.1.bias"] = sd2["skip1.1.bias"] + 2.0e5
    a.net.load_state_dict(sd2); b.net.load_state_dict(sd2)
    img = fixtures.texture_images(2, 96, 128, seed=5).cuda()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ra = a.detectAndCompute(img, top_k=512)
    rb = b.detectAndCompute(img, top_k=512)
    assert any("fp16-pair" in str(m.message) for m in w)
    assert a.net._options.get("fx") == 0 and a.net._effective_option("block1") == 5
    for key, want in ((b"fx", 0), (b"block1", 5)):      # ... and the live handle agrees
        assert lib.xfh_get_option(a.net.handle(), key, C.byref(v)) == 0 and v.value == want, (key, v.value)
    for u, v_ in zip(ra, rb):
        assert torch.equal(u["keypoints"], v_["keypoints"]) and torch.equal(u["scores"], v_["scores"]) and torch.equal(u["descriptors"], v_["descriptors"])
    # (c) tiny activations (fp16 subnormals of the high part): the pair still carries them -- same accuracy as the f32-MFMA kernel
    xs = torch.relu(torch.randn(2, 64, 24, 32, device="cuda")) * 1.0e-6
    ws = a.net.state_dict()
    c = CONV_INDEX["block_fusion.0"]
    ya, yb = torch.empty_like(y), torch.empty_like(y)
    assert lib.xfh_conv_layer(a.net.handle(), c, C.c_void_p(xs.data_ptr()), 2, 24, 32, C.c_void_p(ya.data_ptr()), V_FX, None) == 0
    assert lib.xfh_conv_layer(a.net.handle(), c, C.c_void_p(xs.data_ptr()), 2, 24, 32, C.c_void_p(yb.data_ptr()), V_GENERIC, None) == 0
    assert float((ya - yb).abs().max()) <= 1e-6 * float(yb.abs().max()) + 1e-12      # (against the fp32 direct kernel: the f32-MFMA kernel sums K in another order, a few ulps of the bias)


def test_uint8_ingest_is_bit_identical_to_host_conversion(xf):
    """xfh_backbone_u8: uint8 pixels, NCHW tensor (divisor 1, `.float()`) and numpy HWC image (divisor 255,
    parse_input) against the fp32 entry fed with the host-converted image: bit-identical network outputs and key-points."""
    rs = np.random.RandomState(5)
    img = (fixtures.texture_images(3, 96, 128, seed=21) * 255).round().clamp(0, 255).to(torch.uint8)      # (3,3,96,128) u8
    # (a) uint8 tensor: the reference calls x.float()
    f_a, l_a, h_a, r_a = xf.net.backbone(img.cuda(), want_logits=True, want_heat=True)
    f_b, l_b, h_b, r_b = xf.net.backbone(img.float().cuda(), want_logits=True, want_heat=True)
    for u, v in ((f_a, f_b), (l_a, l_b), (h_a, h_b), (r_a, r_b)):
        assert torch.equal(u, v)
    # (b) numpy HWC uint8 through match_xfeat's parse path vs the public parse_input (/255 on the host)
    hwc = np.ascontiguousarray(img[0].permute(1, 2, 0).numpy())
    fast = xf._parse_input_fast(hwc)
    slow = xf.parse_input(hwc)
    o_fast = xf.detectAndCompute(fast, top_k=512)[0]
    o_slow = xf.detectAndCompute(slow, top_k=512)[0]
    for k in ("keypoints", "scores", "descriptors"):
        assert torch.equal(o_fast[k], o_slow[k]), k
    # (c) gray-scale (H,W) numpy image handed to detectAndCompute (no /255 there, modules/xfeat.py:221-225)
    g = rs.randint(0, 256, size=(64, 96)).astype(np.uint8)
    o_u8 = xf.detectAndCompute(g, top_k=256)[0]
    o_f = xf.detectAndCompute(torch.from_numpy(g)[None, None].float(), top_k=256)[0]
    for k in ("keypoints", "scores", "descriptors"):
        assert torch.equal(o_u8[k], o_f[k]), k


def test_batched_pair_runner_equals_pairwise_match_xfeat(xf):
    """accelerated_features_amd.batching.match_pairs: mixed sizes / dtypes in one list, same results and order
    as XFeat.match_xfeat pair by pair."""
    from accelerated_features_amd.batching import match_pairs
    rs = np.random.RandomState(11)
    def u8(h, w, seed):
        t = fixtures.texture_images(2, h, w, seed=seed)
        return [np.ascontiguousarray((t[i] * 255).round().clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()) for i in range(2)]
    pairs = []
    for seed, (h, w) in enumerate([(96, 128), (64, 96), (96, 128), (96, 128), (64, 96)]):
        a, b = u8(h, w, 30 + seed)
        pairs.append((a, b))
    a, _ = u8(96, 128, 77); _, b = u8(64, 96, 78)
    pairs.append((a, b))                                                  # the two images of this pair differ in size
    pairs.append((fixtures.texture_images(1, 64, 64, seed=5)[0], fixtures.texture_images(1, 64, 64, seed=6)[0]))   # float tensors
    got = match_pairs(xf, pairs, top_k=512, max_pairs=2)
    assert len(got) == len(pairs)
    for (m0, m1), (a, b) in zip(got, pairs):
        if isinstance(a, torch.Tensor):
            r0, r1 = xf.match_xfeat(a[None], b[None], top_k=512)
        else:
            r0, r1 = xf.match_xfeat(a, b, top_k=512)
        assert m0.shape == r0.shape and np.array_equal(m0, r0) and np.array_equal(m1, r1)
    # sharded: rank 1 of 2 gets the second half, same values
    half = match_pairs(xf, pairs, top_k=512, max_pairs=2, rank=1, world=2)
    lo = len(pairs) - len(half)
    for (m0, m1), (r0, r1) in zip(half, got[lo:]):
        assert np.array_equal(m0, r0) and np.array_equal(m1, r1)


def test_batched_star_runner_equals_pairwise_match_xfeat_star(xf):
    """batching.match_pairs_star: size-grouped batches of the semi-dense matcher == match_xfeat_star pair by pair."""
    from accelerated_features_amd.batching import match_pairs_star
    pairs = []
    for seed, (h, w) in enumerate([(160, 192), (128, 160), (160, 192), (160, 192)]):
        t = fixtures.texture_images(2, h, w, seed=50 + seed)
        pairs.append((t[0], torch.roll(t[0], (3, 5), (1, 2)) + 0.01 * t[1]))
    big = fixtures.texture_images(1, 160, 192, seed=60)[0]
    pairs.append((big, big[:, 16:144, 16:176].contiguous()))                  # sizes differ: a crop of the same content
    got = match_pairs_star(xf, pairs, top_k=1024, max_pairs=2)
    assert len(got) == len(pairs)
    for (m0, m1), (a, b) in zip(got, pairs):
        r0, r1 = xf.match_xfeat_star(a[None], b[None], top_k=1024)
        assert m0.shape == r0.shape and np.array_equal(m0, r0) and np.array_equal(m1, r1)
    # the pair whose images differ in size (different numbers of dense key-points per set) against the oracle:
    # rows are (x0,y0,x1,y1); every GPU row must be an oracle row up to 2e-3 px, counts equal up to threshold ties
    a, b = pairs[-1]
    ref = O.match_xfeat_star(fixtures.synthetic_state_dict(0), a[None], b[None], top_k=1024)[0].numpy()
    mine = np.concatenate(got[-1], 1)
    assert abs(len(ref) - len(mine)) <= 2 and len(ref) >= 1      # (synthetic fine_matcher weights keep few rows)
    hits = sum(bool((np.abs(ref - row).max(1) < 2e-3).any()) for row in mine)
    assert hits >= len(mine) - 2, (hits, len(mine))


def test_hipgraph_captured_pipeline_equals_eager(xf):
    """accelerated_features_amd.graphs.CapturedSparsePipeline: one hipGraph replay per call, same kernels, bit-identical
    key-points / descriptors / matches; replays stay correct when the input changes."""
    from accelerated_features_amd.graphs import CapturedSparsePipeline
    pipe = CapturedSparsePipeline(xf, batch=2, height=96, width=128, top_k=512, match=True)
    for seed in (41, 42, 43):
        x = fixtures.texture_images(2, 96, 128, seed=seed).cuda()
        o = pipe(x)
        e = xf.detectAndCompute(x, top_k=512)
        for b in range(2):
            n = o['n_valid'][b]
            assert n == e[b]['keypoints'].shape[0]
            assert torch.equal(o['keypoints'][b, :n], e[b]['keypoints']) and torch.equal(o['scores'][b, :n], e[b]['scores'])
            assert torch.equal(o['descriptors'][b, :n], e[b]['descriptors'])
        i0, i1 = xf.match(e[0]['descriptors'], e[1]['descriptors'], min_cossim=-1)
        assert torch.equal(o['matches'][0][0], i0) and torch.equal(o['matches'][0][1], i1)


def test_fused_two_stage_resize_backbone_is_bit_identical_to_materialised_resizes(xf):
    """xfh_backbone_resized (extract_dualscale's F.interpolate + preprocess_tensor resize + network, modules/xfeat.py:379-381,
    234-238) against xfh_resize_bilinear x2 + xfh_backbone: same gray plane => same features, bit for bit."""
    from accelerated_features_amd.xfeat import _LazyResize
    x = fixtures.texture_images(2, 200, 328, seed=11).cuda()
    for s in (0.6, 1.3, 1.0):
        Hm, Wm = int(np.floor(200 * s)), int(np.floor(328 * s))
        s1 = np.float32(1.0 / s)
        lazy, rh, rw = xf.preprocess_tensor(_LazyResize(x, Hm, Wm, s1, s1))
        assert isinstance(lazy, _LazyResize) and lazy.shape[2] % 32 == 0 and lazy.shape[3] % 32 == 0
        f1, l1, _, r1 = xf.net.backbone(lazy)
        mat, rh2, rw2 = xf.preprocess_tensor(xf._resize(x, Hm, Wm, s1, s1))
        assert (rh, rw) == (rh2, rw2) and tuple(mat.shape) == tuple(lazy.shape)
        f2, l2, _, r2 = xf.net.backbone(mat)
        assert torch.equal(f1, f2) and torch.equal(l1, l2) and torch.equal(r1, r2), s
    # the dense entry point takes the fused route and still matches the oracle (covered by the dense parity test above);
    # here: both scales of a ragged size agree with the per-scale materialised extraction
    mk, sc, ft = xf.extract_dualscale(x, 1000)
    outs = []
    for s, frac in ((0.6, 0.20), (1.3, 0.80)):
        xs = xf._resize(x, int(np.floor(200 * s)), int(np.floor(328 * s)), np.float32(1.0 / s), np.float32(1.0 / s))
        outs.append(xf.extractDense(xs, int(1000 * frac), _scale_div=s))
    assert torch.equal(mk, torch.cat([outs[0][0], outs[1][0]], 1)) and torch.equal(ft, torch.cat([outs[0][1], outs[1][1]], 1))
    # both forms of the fused kernel (option resize2: 1 = the tile's input region staged in LDS, the default; 0 = four-byte gathers) give the same bits
    try:
        xf.net.set_option('resize2', 0)
        mk0, sc0, ft0 = xf.extract_dualscale(x, 1000)
    finally:
        xf.net.set_option('resize2', None)
    assert torch.equal(mk, mk0) and torch.equal(sc, sc0) and torch.equal(ft, ft0)


def test_reference_minimal_example_runs_unchanged():
    """examples/minimal_example.py = the reference's minimal_example.py with only the import changed."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "minimal_example.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "keypoints:  torch.Size([" in r.stdout and "descriptors:  torch.Size([" in r.stdout and "match_xfeat:" in r.stdout
    assert "torch.Size([" in r.stdout.strip().splitlines()[-1] and ", 4])" in r.stdout.strip().splitlines()[-1]


def test_large_frame_2160x3840_vs_oracle(xf, sd):
    """A 4K frame (8.3 MP: 27x the VGA planes, preprocess_tensor resizes 2160 -> 2144) through the sparse path: index
    arithmetic, workspace carving and the NMS capacity logic at a size far from the benchmark configuration."""
    x = fixtures.texture_images(1, 2160, 3840, seed=21)
    out = xf.detectAndCompute(x.cuda(), top_k=4096)[0]
    ref, st = O.detect_and_compute(sd, x, top_k=4096, keep=True)
    rep = parity.compare_keypoints(out, ref[0], heat=st["heat"][0, 0], rw=1.0, rh=2160 / 2144)
    print("4K", rep)
    assert rep["n_test"] == 4096
    kp = out["keypoints"].cpu().numpy()
    assert kp[:, 0].max() <= 3840 and kp[:, 1].max() <= 2160 and kp.min() >= 0


# ----------------------------------------------------------------------------------------------
# reference surface: helper modules / attributes (boundary nits)
# ----------------------------------------------------------------------------------------------
def test_interpolator_attribute_is_the_reference_module(xf, sd):
    """XFeat.interpolator = InterpolateSparse2d('bicubic') like modules/xfeat.py:37; all three modes against the oracle's explicit
    samplers (pinned to F.grid_sample in tests/test_oracle_sampling.py) and against torch's grid_sample on the host."""
    from accelerated_features_amd.interpolator import InterpolateSparse2d
    assert isinstance(xf.interpolator, InterpolateSparse2d) and xf.interpolator.mode == 'bicubic'
    g = torch.Generator().manual_seed(4)
    H, W = 96, 128
    for (C_, Hm, Wm) in ((64, 12, 16), (1, 96, 128), (3, 12, 16)):
        m = torch.randn(2, C_, Hm, Wm, generator=g)
        pos = torch.stack([torch.randint(0, W, (2, 300), generator=g), torch.randint(0, H, (2, 300), generator=g)], -1)
        pos[0, 0] = torch.tensor([0, 0]); pos[0, 1] = torch.tensor([W - 1, H - 1]); pos[0, 2] = torch.tensor([W - 2, H - 2])
        for mode, fn in (("nearest", O.sample_nearest), ("bilinear", O.sample_bilinear), ("bicubic", O.sample_bicubic)):
            mod = InterpolateSparse2d(mode)
            got = mod(m.cuda(), pos.cuda(), H, W).cpu()                     # int64 positions, as detectAndCompute passes them
            assert got.shape == (2, 300, C_)
            for b in range(2):
                parity.assert_close(got[b], fn(m[b], pos[b], H, W), 5e-6, f"{mode} vs oracle")
            grid = (2. * (pos / torch.tensor([W - 1, H - 1])) - 1.).unsqueeze(-2)
            ref = torch.nn.functional.grid_sample(m, grid, mode=mode, align_corners=False).permute(0, 2, 3, 1).squeeze(-2)
            parity.assert_close(got, ref, 1e-5, f"{mode} vs grid_sample")
            fpos = pos.float() + 0.37                                           # fractional positions
            got = mod(m.cuda(), fpos.cuda(), H, W).cpu()
            grid = (2. * (fpos / torch.tensor([W - 1., H - 1.])) - 1.).unsqueeze(-2)
            ref = torch.nn.functional.grid_sample(m, grid, mode=mode, align_corners=False).permute(0, 2, 3, 1).squeeze(-2)
            if mode != "nearest":                                               # (nearest at x.5 ties: compared on integers above)
                parity.assert_close(got, ref, 1e-5, f"{mode} fractional vs grid_sample")
    assert isinstance(xf.kornia_available, bool)


def test_nms_any_odd_kernel_size(xf, sd):
    x = fixtures.texture_images(2, 96, 128, seed=11)
    _, logits, _ = O.backbone(sd, x)
    heat = O.kpts_heatmap(logits)
    for ks in (1, 3, 5, 7, 9):
        ref = O.pad_keypoints(O.nms(heat, 0.02, ks))
        got = xf.NMS(heat.cuda(), threshold=0.02, kernel_size=ks).cpu()
        assert got.shape == ref.shape and torch.equal(got, ref), ks
    with pytest.raises(RuntimeError):
        xf.NMS(heat.cuda(), kernel_size=4)
    # odd widths (the tiled 5x5 kernel loads column pairs and must not see them), incl. maxima on the last column / row
    for Hh, Ww in ((33, 67), (40, 129), (17, 63)):
        g = torch.Generator().manual_seed(Hh * Ww)
        hm = torch.rand(2, 1, Hh, Ww, generator=g)
        hm[0, 0, 5, Ww - 1] = 2.0; hm[1, 0, Hh - 1, Ww - 1] = 3.0; hm[1, 0, 0, 0] = 2.5
        for ks in (3, 5):
            ref = O.pad_keypoints(O.nms(hm, 0.5, ks))
            got = xf.NMS(hm.cuda(), threshold=0.5, kernel_size=ks).cpu()
            assert got.shape == ref.shape and torch.equal(got, ref), (Hh, Ww, ks)


def test_state_dict_reload_and_cpu_inputs(xf, sd):
    """XFeat.load_state_dict (the parent module) must invalidate the packed device weights; CPU tensors handed to the bare network
    raise instead of faulting the GPU."""
    from accelerated_features_amd import XFeat
    from accelerated_features_amd._lib import XFeatHipError
    m = XFeat(weights=sd, top_k=256)
    x = fixtures.texture_images(1, 96, 128, seed=3).cuda()
    a = m.detectAndCompute(x)[0]
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["block_fusion.2.bias"] = sd2["block_fusion.2.bias"] + 0.25
    m.load_state_dict({"net." + k: v for k, v in sd2.items()})          # through the PARENT module
    b = m.detectAndCompute(x)[0]
    ref = XFeat(weights=sd2, top_k=256).detectAndCompute(x)[0]
    assert not torch.equal(a["descriptors"], b["descriptors"])
    assert torch.equal(b["descriptors"], ref["descriptors"]) and torch.equal(b["keypoints"], ref["keypoints"])
    with pytest.raises(XFeatHipError):
        m.net(x.cpu())
    with pytest.raises(XFeatHipError):
        m.net.fine_matcher(torch.zeros(4, 128))


def test_hipgraph_survives_workspace_growth_and_weight_reload_of_the_user_model(sd):
    """The captured graph must not reference memory the user's XFeat can reallocate or free (graphs.py): eager calls with
    larger shapes / a forced larger NMS capacity / load_state_dict on the user's model, then replay."""
    from accelerated_features_amd import XFeat
    from accelerated_features_amd.graphs import CapturedSparsePipeline
    m = XFeat(weights=sd, top_k=512)
    pipe = CapturedSparsePipeline(m, batch=2, height=96, width=128, top_k=512, match=True)
    x = fixtures.texture_images(2, 96, 128, seed=41).cuda()
    first = pipe(x)
    k0, d0 = first['keypoints'].clone(), first['descriptors'].clone()
    m.detectAndCompute(fixtures.texture_images(3, 480, 640, seed=5).cuda())          # user's workspaces grow
    m._detect_device(x, 512, 0.05, cap=96 * 128)                                      # larger NMS capacity
    junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(8)]    # recycle whatever the allocator got back
    m.load_state_dict({"net." + k: v for k, v in sd.items()})                         # user's weight blob freed and rebuilt
    m.detectAndCompute(x)
    again = pipe(x)
    assert torch.equal(again['keypoints'], k0) and torch.equal(again['descriptors'], d0)
    e = m.detectAndCompute(x, top_k=512)
    for b in range(2):
        n = again['n_valid'][b]
        assert torch.equal(again['keypoints'][b, :n], e[b]['keypoints']) and torch.equal(again['descriptors'][b, :n], e[b]['descriptors'])
    del junk


def test_match_filter_and_refine_equals_exact_kernel(xf):
    """xfh_match_mnn's shipped path (fp16 MFMA filter with a derived error window + exact fp32 refine of the flagged 32-wide blocks,
    k_match_f16.hip) against the exact f32 MFMA kernel (option match_exact) on identical inputs: identical index lists -- unit descriptors,
    raw dense features of magnitude ~10, ragged sizes, exact duplicate rows / columns (ties -> lowest index), a similarity cut, and
    degenerate inputs (all-equal / all-zero descriptors: every block is flagged)."""
    g = torch.Generator().manual_seed(12)

    def both(d1, d2, mc):
        a = xf.match(d1.cuda(), d2.cuda(), min_cossim=mc)
        xf.set_option("match_exact", 1)
        try:
            b = xf.match(d1.cuda(), d2.cuda(), min_cossim=mc)
        finally:
            xf.set_option("match_exact", 0)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (d1.shape, d2.shape, mc, len(a[0]), len(b[0]))
        for form in (1, 2):          # the filter's sweep in its two-orientation and its one-orientation form (option match_sweep; the default chooses by shape)
            xf.set_option("match_sweep", form)
            try:
                c = xf.match(d1.cuda(), d2.cuda(), min_cossim=mc)
            finally:
                xf.set_option("match_sweep", 0)
            assert torch.equal(c[0], b[0]) and torch.equal(c[1], b[1]), ("match_sweep", form, d1.shape, d2.shape, mc, len(c[0]), len(b[0]))
        return a

    for n1, n2 in ((4096, 4096), (1000, 31), (33, 257), (1, 1), (2500, 4000), (5000, 300)):
        d1 = torch.nn.functional.normalize(torch.randn(n1, 64, generator=g), dim=-1)
        d2 = torch.nn.functional.normalize(torch.randn(n2, 64, generator=g), dim=-1)
        m = min(n1, n2) // 2
        d2[:m] = torch.nn.functional.normalize(d1[:m] + 0.2 * torch.randn(m, 64, generator=g), dim=-1)     # genuine matches
        if n1 >= 300 and n2 >= 300:
            d2[7] = d1[5]; d2[8] = d1[5]; d1[100] = d1[101]
        for mc in (-1, 0.5):
            i0, _ = both(d1, d2, mc)
        assert len(i0) >= m // 4 or n1 < 50
        both(d1 * 11.0, d2 * 7.0 + 0.3, -1)                                   # raw (un-normalised, shifted) features
    # degenerate: every descriptor identical -> all n2 columns tie in every row (every block flagged, still exact)
    same = torch.nn.functional.normalize(torch.ones(600, 64), dim=-1)
    i0, i1 = both(same, same.clone(), -1)
    assert i0.tolist() == [0] and i1.tolist() == [0]
    z = torch.zeros(300, 64)
    both(z, torch.randn(200, 64, generator=g), -1)
    # batched, device-side counts (the bench path), against per-pair calls
    d = torch.nn.functional.normalize(torch.randn(16, 700, 64, generator=g), dim=-1).cuda()
    nv = torch.tensor([700, 650, 1, 700, 0, 700, 333, 700] * 2, dtype=torch.int32, device="cuda")
    i0, i1, nm = xf.match_pairs_device(d, nv, -1)
    for p in range(8):
        a, b = int(nv[2 * p]), int(nv[2 * p + 1])
        s0, s1 = xf.match(d[2 * p, :a], d[2 * p + 1, :b], min_cossim=-1)
        assert int(nm[p]) == len(s0) and torch.equal(i0[p, :len(s0)], s0) and torch.equal(i1[p, :len(s0)], s1), p


def test_invnorm_by_product_of_the_reliability_head(xf):
    """xfh_backbone's optional invnorm output (reliability head by-product) against 1/max(||feats||, 1e-12) in torch, and the sparse
    results with it handed to xfh_detect_sparse against the internal invnorm pass (NULL): same key-points, descriptors within 1e-6."""
    x = fixtures.texture_images(3, 96, 160, seed=29).cuda()
    feats, _, heat, rel, inv = xf.net.backbone(x, want_logits=False, want_heat=True, want_invnorm=True)
    ref = 1.0 / feats.norm(dim=-1).clamp_min(1e-12)
    assert inv.shape == ref.shape and float(((inv - ref).abs() / ref).max()) <= 1e-6
    B, H, W = 3, 96, 160
    a = xf._detect_call(feats, heat, rel, B, H, W, 0.05, 300, H * W // 8, 1.0, 1.0, inv)
    b = xf._detect_call(feats, heat, rel, B, H, W, 0.05, 300, H * W // 8, 1.0, 1.0, None)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3])
    assert float((a[2] - b[2]).abs().max()) <= 1e-6


def test_sizes_beyond_the_round1_limits(xf, sd):
    """top_k, dense k and descriptor-set sizes above 16384 (round 1 refused them; the reference has no such limits): the run-sorted top-k
    with global-memory rank searches, extractDense(top_k < 1) = every cell of a 1056 x 1312 image (21 648 cells), a 20 000-row match."""
    x = fixtures.texture_images(1, 1056, 1312, seed=33)
    out = xf.detectAndCompute(x.cuda(), top_k=20000, detection_threshold=0.01)[0]
    ref, st = O.detect_and_compute(sd, x, top_k=20000, detection_threshold=0.01, keep=True)
    rep = parity.compare_keypoints(out, ref[0], heat=st["heat"][0, 0], thr=0.01)
    print("top_k 20000:", rep)
    assert rep["n_ref"] > 16384, rep["n_ref"]
    mk, ft = xf.extractDense(x.cuda(), top_k=-1)
    rk, rf, _ = O.extract_dense(sd, x, -1)
    assert mk.shape == rk.shape and mk.shape[1] == 132 * 164 > 16384
    a = {tuple(p): i for i, p in enumerate(mk[0].cpu().numpy().tolist())}
    b = {tuple(p): i for i, p in enumerate(rk[0].numpy().tolist())}
    assert set(a) == set(b)
    ia = torch.tensor([a[k] for k in b]); ib = torch.tensor([b[k] for k in b])
    parity.assert_close(ft[0].cpu()[ia], rf[0][ib], 2e-4, "dense features, all cells")
    g = torch.Generator().manual_seed(8)
    d1 = torch.nn.functional.normalize(torch.randn(300, 64, generator=g), dim=-1)
    d2 = torch.nn.functional.normalize(torch.randn(20000, 64, generator=g), dim=-1)
    d2[17000] = d1[5]; d2[123] = d1[7]
    for a_, b_ in ((d1, d2), (d2, d1)):
        i0, i1 = xf.match(a_.cuda(), b_.cuda(), min_cossim=-1)
        o0, o1 = O.match_mnn(a_, b_, -1)
        assert torch.equal(i0.cpu(), o0) and torch.equal(i1.cpu(), o1)


def test_match_with_f16_copies_from_the_descriptor_kernel(xf):
    """xfh_detect_sparse's optional desc_f16 output handed to xfh_match_mnn (no conversion passes inside the matcher): bit pattern = fp16 RNE of
    256 * the fp32 descriptors, match lists identical to the plain call."""
    x = fixtures.texture_images(4, 192, 256, seed=61).cuda()
    kp, sc, de, nv, nc, cap, hw, d16 = xf._detect_device(x, 1024, 0.05, want_f16=True)
    ref16 = (de * 256.0).to(torch.float16)
    assert d16.dtype == torch.float16 and torch.equal(d16.view(torch.int16), ref16.view(torch.int16))
    a = xf.match_pairs_device(de, nv, -1)
    b = xf.match_pairs_device(de, nv, -1, d16)
    c = xf.match_pairs_device(de, nv, 0.5, d16)
    d = xf.match_pairs_device(de, nv, 0.5)
    for u, v in ((a, b), (c, d)):
        assert torch.equal(u[2], v[2])
        for p in range(2):
            n = int(u[2][p])
            assert n > 50 and torch.equal(u[0][p, :n], v[0][p, :n]) and torch.equal(u[1][p, :n], v[1][p, :n])


def test_match_on_rounding_aligned_adversarial_sets(xf):
    """The filter window against descriptor sets whose fp16 rounding errors are ALIGNED (tests/adversarial.py: components a hair below / above
    rounding midpoints, the true best match losing 1.9u in the rounded product to a competitor in another 32-wide block -- half the window loses
    every one of them, tests/test_match_filter_window.py).  Default path == exact f32-MFMA kernel == oracle, through xfh_match_mnn with its own
    conversion passes and -- for the unit-norm sets -- with caller-provided fp16 copies (the bench path), both directions, with and without a cut."""
    import adversarial as A
    total = 0
    for name, d1, d2, unit in A.sets():
        t1, t2 = A.as_torch(d1), A.as_torch(d2)
        o0, o1 = O.match_mnn(t1, t2, -1)
        n_strict = A.check_mnn_fp64(d1, d2, o0.numpy(), o1.numpy())             # the oracle itself against float64 (tie allowance 2e-6)
        for mc in (-1, 0.3):
            a = xf.match(t1.cuda(), t2.cuda(), min_cossim=mc)
            xf.set_option("match_exact", 1)
            try:
                b = xf.match(t1.cuda(), t2.cuda(), min_cossim=mc)
            finally:
                xf.set_option("match_exact", 0)
            forms = []
            for form in (1, 2):      # both forms of the filter's sweep (option match_sweep)
                xf.set_option("match_sweep", form)
                try:
                    forms.append(xf.match(t1.cuda(), t2.cuda(), min_cossim=mc))
                finally:
                    xf.set_option("match_sweep", 0)
            for got in (a, b) + tuple(forms):
                A.check_mnn_fp64(d1, d2, got[0].cpu().numpy(), got[1].cpu().numpy(), mc)
            if "all_equal" not in name:
                for c in forms:
                    assert torch.equal(c[0], b[0]) and torch.equal(c[1], b[1]), (name, mc, "match_sweep")
            if "all_equal" not in name:               # (all-equal rows: every pair is a tie, any consistent answer passes the check above)
                assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (name, mc, len(a[0]), len(b[0]))
                if mc < 0:
                    assert torch.equal(a[0].cpu(), o0) and torch.equal(a[1].cpu(), o1), (name, len(a[0]), len(o0))
            total += len(a[0])
        if unit:
            # caller-provided copies: one pair (frames 0 and 1 of a "batch"), rows padded to a common capacity with device-side counts
            K = max(len(d1), len(d2))
            de = torch.zeros(2, K, 64); de[0, :len(d1)] = t1; de[1, :len(d2)] = t2
            de = de.cuda()
            nv = torch.tensor([len(d1), len(d2)], dtype=torch.int32, device="cuda")
            d16 = (de * 256.0).to(torch.float16)
            i0, i1, nm = xf.match_pairs_device(de, nv, -1, d16)
            n = int(nm[0])
            A.check_mnn_fp64(d1, d2, i0[0, :n].cpu().numpy(), i1[0, :n].cpu().numpy())
            if "all_equal" not in name:
                assert n == len(o0) and torch.equal(i0[0, :n].cpu(), o0) and torch.equal(i1[0, :n].cpu(), o1), name
        print(name, d1.shape, d2.shape, "strict mutual matches", n_strict, "reported", len(o0))
    assert total > 1500


@pytest.mark.parametrize("concurrent", [False, True])
def test_frame_stream_lanes_deliver_the_synchronous_results_in_order(xf, sd, concurrent):
    """accelerated_features_amd.streaming.FrameStream: batches on two lanes (two handles, asynchronous read-back of the ragged counts; one HIP stream, or one per
    lane) give, ticket by ticket, exactly what the synchronous device path gives for the same batch -- key-points, scores, descriptors, match lists, counts."""
    from accelerated_features_amd.streaming import FrameStream
    from accelerated_features_amd import XFeat
    fs = FrameStream(weights=sd, top_k=512, lanes=2, concurrent=concurrent)
    xs_ = XFeat(weights=sd, top_k=512)                                        # (the lanes run the library's default kernel mix in both modes)
    batches = [fixtures.texture_images(4, 96, 128, seed=40 + i).cuda() for i in range(5)]
    batches[3] = fixtures.texture_images(6, 64, 96, seed=99).cuda()          # another shape / batch size in the middle of the stream
    want = []
    for x in batches:
        kp, sc, de, nv, nc, cap, hw, d16 = xs_._detect_device(x, 512, 0.05, want_f16=True)
        i0, i1, nm = xs_.match_pairs_device(de, nv, -1, d16)
        want.append((kp.clone(), sc.clone(), de.clone(), nv.cpu(), nc.cpu(), i0.clone(), i1.clone(), nm.cpu()))
    got, tickets = [], []
    for i, x in enumerate(batches):
        if fs.in_flight == fs.lanes:
            got.append(fs.result(tickets[len(got)]))
        tickets.append(fs.submit(x))
    with pytest.raises(RuntimeError):
        fs.result(tickets[-1])                                               # tickets retire in order
    got += fs.drain()
    assert [g["ticket"] for g in got] == tickets == list(range(5)) and fs.in_flight == 0
    for g, w in zip(got, want):
        kp, sc, de, nv, nc, i0, i1, nm = w
        assert torch.equal(g["n_valid"], nv) and torch.equal(g["n_candidates"], nc) and torch.equal(g["n_matches"], nm)
        for b in range(len(nv)):
            n = int(nv[b])
            assert torch.equal(g["keypoints"][b, :n], kp[b, :n]) and torch.equal(g["scores"][b, :n], sc[b, :n]) and torch.equal(g["descriptors"][b, :n], de[b, :n])
        for p_ in range(len(nm)):
            n = int(nm[p_])
            assert n > 0 and torch.equal(g["idx0"][p_, :n], i0[p_, :n]) and torch.equal(g["idx1"][p_, :n], i1[p_, :n])
    with pytest.raises(RuntimeError):
        fs.result()
    t = [fs.submit(batches[0]), fs.submit(batches[1])]
    with pytest.raises(RuntimeError):
        fs.submit(batches[2])                                                # both lanes busy
    fs.drain()
    with pytest.raises(ValueError):
        fs.submit(batches[0][:3])


@pytest.mark.parametrize("concurrent", [False, True])
def test_frame_stream_at_the_bench_shape_is_bit_stable_over_many_steps(xf, sd, concurrent):
    """Two lanes at BASELINE configs[1] (64 VGA frames, top_k 4096): 40 batches in flight two at a time -- every retired result equals the synchronous one
    bit for bit (one stream: the synchronous kernel order; a stream per lane: two handles share nothing but the device)."""
    from accelerated_features_amd.streaming import FrameStream
    from accelerated_features_amd import XFeat
    x = torch.cat([fixtures.texture_images(8, 480, 640, seed=77)] * 8).cuda()
    xs_ = XFeat(weights=sd, top_k=4096)

    def sync_result():
        kp, sc, de, nv, nc, cap, hw, d16 = xs_._detect_device(x, 4096, 0.05, want_f16=True)
        i0, i1, nm = xs_.match_pairs_device(de, nv, -1, d16)
        return kp, sc, de, nv.cpu(), nm.cpu(), i0, i1
    kp0, sc0, de0, nv_h, nm_h, i00, i10 = sync_result()
    fs = FrameStream(weights=sd, top_k=4096, lanes=2, concurrent=concurrent)
    got = []
    for step in range(40):
        if fs.in_flight == fs.lanes:
            got.append(fs.result())
        fs.submit(x)
    got += fs.drain()
    assert len(got) == 40
    bad = []
    for t, r in enumerate(got):
        what = []
        if not (torch.equal(r["n_valid"], nv_h) and torch.equal(r["n_matches"], nm_h)): what.append("counts")
        for name, ref_t in (("keypoints", kp0), ("scores", sc0), ("descriptors", de0)):
            if not torch.equal(r[name], ref_t):
                what.append(f"{name} of images {sorted(set((r[name] != ref_t).flatten(1).any(1).nonzero().flatten().tolist()))}")
        if not what:
            for p in (0, 13, 31):
                n = int(nm_h[p])
                if not (torch.equal(r["idx0"][p, :n], i00[p, :n]) and torch.equal(r["idx1"][p, :n], i10[p, :n])): what.append(f"match list {p}")
        if what:
            bad.append((t, what))
    if bad:                                                                  # which side moved?  the synchronous result once more
        kp1, sc1, de1, nv1, nm1, i01, i11 = sync_result()
        ref_stable = torch.equal(kp1, kp0) and torch.equal(de1, de0) and torch.equal(nv1, nv_h) and torch.equal(nm1, nm_h)
        raise AssertionError(f"{len(bad)} of 40 results differ from the synchronous one (synchronous result reproducible: {ref_stable}): {bad[:6]}")


@pytest.mark.parametrize("opts", [{}, {"fx": 0, "block1": 5}, {"fx": FX_CONV64 | FX_CONV24, "block1": 5}])
def test_two_streams_and_cold_instruction_cache_soak(sd, opts):
    """Time-boxed soak.  The bench-shape backbone + sparse step + match on one HIP stream while a second stream runs foreign kernels (another
    model's backbone = every kernel of this library, a large copy, a rocBLAS GEMM), then the same with every matrix-core
    kernel starting on an invalidated instruction cache (xfh_debug_cold_start: the condition that made round 3's split-bf16 key-point head -- deleted since -- deliver wrong
    16-cell blocks) -- every network output and every match list of every step bit-identical to the quiet, warm reference.  Parametrised over the kernel sets that
    ship: the defaults, the range fallback as a whole, and a mix (f32-MFMA heads and vector block1 next to the fp16-pair convolutions)."""
    import threading
    import time
    from accelerated_features_amd import XFeat
    lib = _lib().load()
    x = torch.cat([fixtures.texture_images(8, 480, 640, seed=77)] * 8).cuda()
    xf_ = XFeat(weights=sd, top_k=4096)
    for k, v in opts.items():
        xf_.set_option(k, v)
    names = ("feats", "heat", "rel", "inv", "kpts", "desc", "idx0", "idx1", "counts")

    def step():
        f, _, h, r, inv = xf_.net.backbone(x, want_logits=False, want_heat=True, want_invnorm=True)
        kp, sc, de, nv, nc, cap, hw, d16 = xf_._detect_device(x, 4096, 0.05, want_f16=True)
        i0, i1, nm = xf_.match_pairs_device(de, nv, -1, d16)
        return f, h, r, inv, kp, de, i0, i1, torch.cat([nv, nc, nm])
    with torch.inference_mode():
        want = [t.clone() for t in step()]
    torch.cuda.synchronize()
    stop = threading.Event()

    def foreign():
        st = torch.cuda.Stream()
        other = XFeat(weights=sd, top_k=4096)
        a = torch.empty(32 << 20, device="cuda"); b = torch.empty_like(a)
        m1 = torch.randn(2048, 2048, device="cuda"); m2 = torch.randn(2048, 2048, device="cuda")
        with torch.cuda.stream(st), torch.inference_mode():
            evs = []
            while not stop.is_set():
                other.net.backbone(x, want_logits=False, want_heat=True, want_invnorm=True)
                b.copy_(a)
                torch.mm(m1, m2)
                ev = torch.cuda.Event(); ev.record(st); evs.append(ev)
                if len(evs) > 4:
                    evs.pop(0).synchronize()
            st.synchronize()

    def soak(seconds):
        bad = torch.zeros(len(names), dtype=torch.int64, device="cuda")
        t0, n = time.time(), 0
        with torch.inference_mode():
            while time.time() - t0 < seconds:
                for _ in range(20):
                    for k, t in enumerate(step()):
                        # (rows of the padded tensors beyond the counts are unspecified: compare what the counts cover -- everything for the network outputs)
                        if k < 4 or k == 8:
                            bad[k] += (t != want[k]).any()
                        else:
                            bad[k] += (t[:, :256] != want[k][:, :256]).any()
                    n += 1
                torch.cuda.synchronize()
        return n, bad.tolist()
    th = threading.Thread(target=foreign, daemon=True)
    th.start()
    time.sleep(0.3)
    try:
        n1, bad1 = soak(8.0)
    finally:
        stop.set(); th.join()
    lib.xfh_debug_cold_start(1)
    try:
        n2, bad2 = soak(6.0)
    finally:
        lib.xfh_debug_cold_start(0)
    print(f"options {opts}: {n1} steps next to foreign kernels: differing {dict(zip(names, bad1))}; {n2} cold-start steps: differing {dict(zip(names, bad2))}")
    assert n1 >= 200 and n2 >= 200
    assert not any(bad1) and not any(bad2), (dict(zip(names, bad1)), dict(zip(names, bad2)))
