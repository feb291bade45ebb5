"""Generate the committed golden vectors by running the UNMODIFIED reference on CPU.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_golden.py

Step 1 calibrates the BatchNorm running statistics of the synthetic fixture (one train-mode
forward of the reference ``XFeatModel`` with ``momentum=None`` on seeded texture images) and
writes ``bn_stats.npz``.  Step 2 loads the resulting state_dict into the reference
``modules.xfeat.XFeat`` (CPU) and stores what its public methods return on seeded inputs.
Everything here is produced by reference code; the oracle and the HIP path are compared
against these files (tests/test_oracle_golden.py, tests/test_gpu_parity.py).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

import fixtures  # noqa: E402
from modules.xfeat import XFeat  # noqa: E402  (the reference)


WHITEN_FLOOR = 0.1          # eigenvalue floor of the descriptor whitening: amplification <= 1/sqrt(0.1) = 3.2x


def _dense_feats(net, xf, imgs, scales=(0.6, 1.0, 1.3)):
    """Raw dense features (n,64) of `imgs` at the scales the semi-dense path works on."""
    out = []
    for s in scales:
        x = torch.nn.functional.interpolate(imgs, scale_factor=s, align_corners=False, mode="bilinear")
        x, _, _ = xf.preprocess_tensor(x)
        f, _, _ = net(x)
        out.append(f.permute(0, 2, 3, 1).reshape(-1, 64))
    return torch.cat(out)


def calibrate():
    """Fixture calibration (all of it with the reference's own modules, CPU):
    1. BatchNorm2d running statistics of the backbone + heads: one train-mode forward (momentum=None).
    2. block_fusion.2 (the plain 1x1 conv that emits the descriptors) is composed with a ZCA whitening of the dense
       features it produces on seeded textures (eigenvalues floored at WHITEN_FLOOR).  Reason: with plain random
       weights the raw descriptors share one dominant direction and have heavy-tailed norms, so the raw-dot-product
       mutual-NN of the semi-dense matcher (xfeat.py:265-290) finds ~1 match per 4095 and
       match_xfeat_star has nothing to refine.  Whitened descriptors behave like a trained net's: a noisy copy of an
       image yields thousands of mutual matches.
    3. BatchNorm1d statistics of fine_matcher on MATCHED pairs of those descriptors (what refine_matches feeds it).
    Everything lands in bn_stats.npz (the conv summation order of the calibrating machine is baked in)."""
    torch.manual_seed(0)
    sd = fixtures.raw_state_dict(0)
    xf = XFeat(weights=sd)
    net = xf.net
    for m in net.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            m.momentum = None
            m.reset_running_stats()
    imgs = fixtures.texture_images(4, 256, 320, seed=123)
    net.train()
    with torch.no_grad():
        net.fine_matcher.eval()
        net(imgs)
    net.eval()
    out = {}
    with torch.no_grad():
        # -- 2. whitening of the descriptor layer
        cal = fixtures.texture_images(2, 512, 512, seed=124)
        f = _dense_feats(net, xf, cal).double()
        mu = f.mean(0)
        ev, V = torch.linalg.eigh(torch.cov(f.T))
        Wh = V @ torch.diag(ev.clamp_min(WHITEN_FLOOR).rsqrt()) @ V.T
        conv = net.block_fusion[2]
        W0, b0 = conv.weight[:, :, 0, 0].double(), conv.bias.double()
        conv.weight.copy_((Wh @ W0).float()[:, :, None, None])
        conv.bias.copy_((Wh @ (b0 - mu)).float())
        out["block_fusion.2.weight"] = conv.weight.numpy().copy()
        out["block_fusion.2.bias"] = conv.bias.numpy().copy()
        print("whitening: eigenvalues %.2e .. %.2e, %d floored" % (float(ev[0]), float(ev[-1]), int((ev < WHITEN_FLOOR).sum())))
        # the heat-map head reads the descriptors: re-calibrate its BatchNorms on the whitened features
        for m in net.heatmap_head.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.reset_running_stats()
                m.train()
        net(imgs)
        net.eval()
        # -- 3. fine matcher on matched pairs: descriptors of an image and of a noisy copy, same cells
        g = torch.Generator().manual_seed(5)
        noisy = cal + 0.01 * torch.randn(cal.shape, generator=g)
        fa, fb = _dense_feats(net, xf, cal), _dense_feats(net, xf, noisy)
        sel = torch.randperm(len(fa), generator=g)[:8192]
        net.fine_matcher.train()
        net.fine_matcher(torch.cat([fa[sel], fb[sel]], -1))
        net.eval()
    for k, v in net.state_dict().items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            out[k] = v.numpy().astype(np.float32)
    np.savez_compressed(fixtures.BN_STATS, **out)
    print("bn_stats:", len(out), "arrays")


def main():
    calibrate()
    torch.set_num_threads(8)
    sd = fixtures.synthetic_state_dict(0)
    xf = XFeat(weights=sd, top_k=4096, detection_threshold=0.05)
    assert str(xf.dev) == "cpu"

    # ---- G1: small batch, every network output (96x128, B=2) ---------------------------
    x = fixtures.texture_images(2, 96, 128, seed=11)
    with torch.no_grad():
        feats, logits, rel = xf.net(x)
        heat = xf.get_kpts_heatmap(logits)
    out = xf.detectAndCompute(x, top_k=256)
    g1 = {"feats": feats.numpy(), "logits": logits.numpy(), "reliability": rel.numpy(), "heat": heat.numpy()}
    for b, o in enumerate(out):
        for k, v in o.items():
            g1[f"{k}{b}"] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "g1_small.npz"), **g1)
    print("g1:", [len(o["keypoints"]) for o in out])

    # ---- G2: VGA pair, top_k=4096: detectAndCompute + match -----------------------------
    a, b = fixtures.shifted_pair(1, 480, 640, seed=7)
    oa = xf.detectAndCompute(a)[0]
    ob = xf.detectAndCompute(b)[0]
    i0, i1 = xf.match(oa["descriptors"], ob["descriptors"], min_cossim=-1)
    j0, j1 = xf.match(oa["descriptors"], ob["descriptors"], min_cossim=0.82)   # the reference's default: (nearly) empty on this fixture
    h0, h1 = xf.match(oa["descriptors"], ob["descriptors"], min_cossim=0.55)   # about half of the mutual matches pass
    s = oa["descriptors"] @ ob["descriptors"].t()
    top2 = torch.topk(s, 2, dim=1)[0]
    g2 = {}
    for tag, o in (("a", oa), ("b", ob)):
        g2[f"kp_{tag}"] = o["keypoints"].numpy().astype(np.int16)
        assert np.array_equal(g2[f"kp_{tag}"].astype(np.float32), o["keypoints"].numpy())
        g2[f"sc_{tag}"] = o["scores"].numpy()
        g2[f"desc_{tag}_every8"] = o["descriptors"][::8].numpy()
        g2[f"desc_{tag}_rowsum"] = o["descriptors"].double().sum(1).numpy()
    g2.update(idx0=i0.numpy().astype(np.int32), idx1=i1.numpy().astype(np.int32),
              idx0_082=j0.numpy().astype(np.int32), idx1_082=j1.numpy().astype(np.int32),
              idx0_055=h0.numpy().astype(np.int32), idx1_055=h1.numpy().astype(np.int32),
              row_gap=(top2[:, 0] - top2[:, 1]).numpy())
    np.savez_compressed(os.path.join(HERE, "g2_vga_pair.npz"), **g2)
    print("g2: kpts", len(oa["keypoints"]), len(ob["keypoints"]), "matches", len(i0), "matches@0.82", len(j0), "matches@0.55", len(h0))

    # ---- G3: non-/32 input through match_xfeat (numpy uint8 HWC, 200x300 -> 192x288) ----
    rs = np.random.RandomState(3)
    ta, tb = fixtures.shifted_pair(1, 200, 300, seed=21, shift=(5, 9))
    ia = (ta[0].permute(1, 2, 0).numpy() * 255).clip(0, 255).astype(np.uint8)
    ib = (tb[0].permute(1, 2, 0).numpy() * 255).clip(0, 255).astype(np.uint8)
    m0, m1 = xf.match_xfeat(ia, ib, top_k=1024)
    np.savez_compressed(os.path.join(HERE, "g3_match_xfeat.npz"), m0=m0, m1=m1)
    print("g3:", m0.shape)

    # ---- G4: semi-dense: detectAndComputeDense + match_xfeat_star (B=2, 160x192) --------
    sa, sb = fixtures.shifted_pair(2, 160, 192, seed=31, shift=(8, 8))
    dd = xf.detectAndComputeDense(sa, top_k=512)
    res = xf.match_xfeat_star(sa, sb, top_k=512)
    # refine with a forced index list so the MLP path is exercised on many rows
    d0 = xf.detectAndComputeDense(sa, top_k=512)
    d1 = xf.detectAndComputeDense(sb, top_k=512)
    n = d0["keypoints"].shape[1]
    g = torch.Generator().manual_seed(9)
    forced = [(torch.arange(n), torch.randperm(n, generator=g)) for _ in range(2)]
    with torch.no_grad():
        ref0 = xf.refine_matches({k: v.clone() for k, v in d0.items()}, d1, forced, 0)
        ref1 = xf.refine_matches({k: v.clone() for k, v in d0.items()}, d1, forced, 1)
    g4 = {"dense_kp": dd["keypoints"].numpy(), "dense_desc": dd["descriptors"].numpy(),
          "dense_scales": dd["scales"].numpy(),
          "star0": res[0].numpy(), "star1": res[1].numpy(),
          "forced_perm0": forced[0][1].numpy(), "forced_perm1": forced[1][1].numpy(),
          "refine0": ref0.numpy(), "refine1": ref1.numpy()}
    np.savez_compressed(os.path.join(HERE, "g4_dense.npz"), **g4)
    print("g4: star", [len(r) for r in res], "forced refine", len(ref0), len(ref1), "of", n)

    # ---- G5: BASELINE configs[0]: assets/ref.png <-> tgt.png (600x800 RGB uint8; preprocess resizes 600 -> 576) ----
    from PIL import Image
    im0 = np.asarray(Image.open("/root/reference/assets/ref.png").convert("RGB"))
    im1 = np.asarray(Image.open("/root/reference/assets/tgt.png").convert("RGB"))
    assert im0.shape == (600, 800, 3) and im1.shape == (600, 800, 3) and im0.dtype == np.uint8
    m0, m1 = xf.match_xfeat(im0, im1, top_k=4096)
    # what minimal_example / the notebooks do: detectAndCompute on the parsed image, then match()
    o0 = xf.detectAndCompute(xf.parse_input(im0), top_k=4096)[0]
    o1 = xf.detectAndCompute(xf.parse_input(im1), top_k=4096)[0]
    i0, i1 = xf.match(o0["descriptors"], o1["descriptors"], min_cossim=0.5)
    s0, s1 = xf.match_xfeat_star(im0, im1, top_k=4096)
    g5 = {"img0": im0, "img1": im1, "m0": m0, "m1": m1, "idx0_050": i0.numpy().astype(np.int32),
          "idx1_050": i1.numpy().astype(np.int32), "star0": s0, "star1": s1}
    for t, o in (("0", o0), ("1", o1)):
        g5[f"kp{t}"] = o["keypoints"].numpy()
        g5[f"sc{t}"] = o["scores"].numpy()
        g5[f"desc{t}_every8"] = o["descriptors"][::8].numpy()
    np.savez_compressed(os.path.join(HERE, "g5_assets.npz"), **g5)
    print("g5: kpts", len(o0["keypoints"]), len(o1["keypoints"]), "match_xfeat", m0.shape, "match@0.5", len(i0), "star", s0.shape)

    # ---- G6: match_xfeat_star with MANY refined rows: noisy copy pair, 320x384, B=2, top_k=2048 ----------------
    sa, sb = fixtures.star_pair(2, 320, 384, seed=41)
    res = xf.match_xfeat_star(sa, sb, top_k=2048)
    d0 = xf.detectAndComputeDense(sa, top_k=2048)
    d1 = xf.detectAndComputeDense(sb, top_k=2048)
    bm = xf.batch_match(d0["descriptors"], d1["descriptors"])
    g6 = {"star0": res[0].numpy(), "star1": res[1].numpy(),
          "kp_a": d0["keypoints"].numpy(), "kp_b": d1["keypoints"].numpy(),
          "desc_a_every8": d0["descriptors"][:, ::8].numpy(),
          "bm0_idx0": bm[0][0].numpy().astype(np.int32), "bm0_idx1": bm[0][1].numpy().astype(np.int32),
          "bm1_idx0": bm[1][0].numpy().astype(np.int32), "bm1_idx1": bm[1][1].numpy().astype(np.int32)}
    np.savez_compressed(os.path.join(HERE, "g6_star.npz"), **g6)
    print("g6: mutual", [len(b[0]) for b in bm], "refined rows", [len(r) for r in res])


if __name__ == "__main__":
    main()
