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


def calibrate():
    torch.manual_seed(0)
    sd = fixtures.raw_state_dict(0)
    xf = XFeat(weights=sd)
    net = xf.net
    for m in net.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            m.momentum = None
            m.reset_running_stats()
    imgs = fixtures.texture_images(4, 256, 320, seed=123)
    # backbone + heads: one train-mode forward accumulates batch statistics
    net.train()
    with torch.no_grad():
        for mod in (net.fine_matcher,):
            mod.eval()
        feats, _, _ = net(imgs)
    net.eval()
    # fine matcher: pairs of raw dense features, as refine_matches feeds it
    with torch.no_grad():
        feats, _, _ = net(imgs)
        f = feats.permute(0, 2, 3, 1).reshape(-1, 64)
        g = torch.Generator().manual_seed(5)
        i0 = torch.randperm(len(f), generator=g)[:4096]
        i1 = torch.randperm(len(f), generator=g)[:4096]
        net.fine_matcher.train()
        net.fine_matcher(torch.cat([f[i0], f[i1]], -1))
        net.eval()
    out = {}
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
    j0, j1 = xf.match(oa["descriptors"], ob["descriptors"], min_cossim=0.82)
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
              row_gap=(top2[:, 0] - top2[:, 1]).numpy())
    np.savez_compressed(os.path.join(HERE, "g2_vga_pair.npz"), **g2)
    print("g2: kpts", len(oa["keypoints"]), len(ob["keypoints"]), "matches", len(i0), "matches@0.82", len(j0))

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


if __name__ == "__main__":
    main()
