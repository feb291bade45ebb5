"""The shipped default kernels END TO END against the reference-made goldens, without a GPU: block1 with block1.2 / block1.3 on the fp16 matrix cores (mode 7), the fp16-pair
heads and -- in the second test -- 16 of the 17 convolution layers run in the host emulation (tests/emu/) on the golden fixtures' images and weights, everything else (NMS, scores,
top-k, descriptors; in the first test block2 .. feats too) is the oracle's fp32 restatement, and the resulting key-point lists are compared with what the UNMODIFIED
reference wrote into tests/golden/ (g1_small: 2 x 256 key-points; g2_vga_pair: 2 x 4096 at VGA) by the GPU suite's own comparator.  The range fallback's forms (mode 5, the
f32-MFMA head) run next to them as the control: the default may not be further from the reference than the fallback."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures
import parity
from oracle import xfeat_oracle as O

CLANG = "/opt/rocm/lib/llvm/bin/clang++"


class Bins(list):
    def __init__(self):
        super().__init__()
        self.by_name = {}


@pytest.fixture(scope="module")
def bins():
    if not os.path.exists(CLANG):
        pytest.skip("no host clang")
    td = tempfile.mkdtemp()
    import test_kernels_emulated as sliced      # (the 24-channel kernels are emulated as slices of k_conv_bx.hip)
    for fname, fn in (("conv_bx24_slice.hpp", sliced._slice_conv_bx24), ("weight_split_slice.hpp", sliced._slice_weight_split), ("bx_split_slice.hpp", sliced._slice_bx_split),
                      ("pyramid_slice.hpp", sliced._slice_pyramid), ("gray_slice.hpp", sliced._slice_gray)):
        open(os.path.join(td, fname), "w").write(fn())
    out = Bins()
    for name in ("block1_emu", "head_emu", "conv_bx24_emu", "conv_bx64s2_emu", "conv_rs64_emu", "pyramid53_emu", "gray_emu"):
        out.append(os.path.join(td, name))
        out.by_name[name] = out[-1]
        subprocess.run([CLANG, "-O1", "-w", "-std=c++20", "-pthread", "-I", td, "-I", os.path.join(ROOT, "accelerated_features_amd", "csrc"), "-I", os.path.join(ROOT, "tests", "emu"),
                        os.path.join(ROOT, "tests", "emu", name + ".cpp"), "-o", out[-1]], check=True)
    return out


def fold(sd, name):
    s = 1.0 / torch.sqrt(sd[f"{name}.layer.1.running_var"].double() + 1e-5)
    w = sd[f"{name}.layer.0.weight"].double()
    return (w * s.view(-1, 1, 1, 1)).float(), (-sd[f"{name}.layer.1.running_mean"].double() * s).float()
def gray_coef(x):
    gray = x.mean(1)
    gd = gray.double()
    alpha = 1.0 / torch.sqrt(gd.var((1, 2), unbiased=False) + 1e-5)
    return gray, torch.stack([alpha, -gd.mean((1, 2)) * alpha], 1).float()
def run_block1(bins, sd, gray, coef, mode):
    B, H, W = gray.shape
    w1, b1 = fold(sd, "block1.0"); w2, b2 = fold(sd, "block1.1"); w3, b3 = fold(sd, "block1.2"); w4, b4 = fold(sd, "block1.3")
    skw, skb = sd["skip1.1.weight"].float(), sd["skip1.1.bias"].float()
    kc = lambda t: t.permute(1, 2, 3, 0).reshape(-1).contiguous()
    pad = lambda t: torch.cat([t.reshape(-1), torch.zeros(32 - t.numel())])
    blob = np.concatenate([np.array([B, H, W, mode], np.int32).view(np.float32)] + [t.numpy().astype(np.float32).reshape(-1) for t in (
        gray, coef, kc(w1), b1, kc(w2), b2, kc(w3), b3, kc(w4), pad(b4), pad(skw), pad(skb))])
    out = subprocess.run([bins[0]], input=blob.tobytes(), capture_output=True, check=True, timeout=3000).stdout
    assert int(np.frombuffer(out[-4:], np.int32)[0]) == 0
    return torch.from_numpy(np.frombuffer(out[:-4], np.float32).reshape(B, 24, H // 4, W // 4).copy())
def run_head(bins, sd, gray, coef, fx):
    B, H, W = gray.shape
    ws, bs = [], []
    for i in range(3):
        w, b = fold(sd, f"keypoint_head.{i}"); ws.append(w.view(64, 64)); bs.append(b)
    ws.append(sd["keypoint_head.3.weight"].view(65, 64).float()); bs.append(sd["keypoint_head.3.bias"].float())
    blob = np.concatenate([np.array([1, fx, B, H, W], np.int32).view(np.float32)] + [np.asarray(a, np.float32).reshape(-1) for a in [gray, coef] + ws + bs]).tobytes()
    out = subprocess.run([bins[1]], input=blob, capture_output=True, check=True, timeout=3000).stdout
    assert int(np.frombuffer(out[-4:], np.int32)[0]) == 0
    return torch.from_numpy(np.frombuffer(out[:4 * B * H * W], np.float32).reshape(B, 1, H, W).copy())
def rest_of_backbone(sd, x1):
    a = O._basic(sd, "block2.0", x1); a = O._basic(sd, "block2.1", a)
    x3 = O._basic(sd, "block3.0", a, 2); x3 = O._basic(sd, "block3.1", x3); x3 = O._basic(sd, "block3.2", x3, 1, 1)
    x4 = O._basic(sd, "block4.0", x3, 2); x4 = O._basic(sd, "block4.1", x4); x4 = O._basic(sd, "block4.2", x4)
    x5 = O._basic(sd, "block5.0", x4, 2); x5 = O._basic(sd, "block5.1", x5); x5 = O._basic(sd, "block5.2", x5); x5 = O._basic(sd, "block5.3", x5, 1, 1)
    hw = tuple(x3.shape[-2:])
    f = x3 + F.interpolate(x4, hw, mode="bilinear") + F.interpolate(x5, hw, mode="bilinear")
    f = O._basic(sd, "block_fusion.0", f); f = O._basic(sd, "block_fusion.1", f)
    feats = O._plain(sd, "block_fusion.2", f)
    h = O._basic(sd, "heatmap_head.0", feats, 1, 1); h = O._basic(sd, "heatmap_head.1", h, 1, 1)
    return feats, torch.sigmoid(O._plain(sd, "heatmap_head.2", h))
def detect(feats, heat, rel, top_k, H, W):
    B = feats.shape[0]
    fn = F.normalize(feats, dim=1)
    cand = O.nms(heat, 0.05, 5); mk = O.pad_keypoints(cand); N = mk.shape[1]
    scores = torch.empty((B, N), dtype=torch.float32)
    for b in range(B):
        scores[b] = O.sample_nearest(heat[b], mk[b], H, W)[:, 0] * O.sample_bilinear(rel[b], mk[b], H, W)[:, 0]
    scores[torch.all(mk == 0, dim=-1)] = -1
    order = torch.argsort(-scores)
    mk = torch.gather(mk, 1, order[..., None].expand(-1, -1, 2))[:, :top_k]
    scores = torch.gather(scores, 1, order)[:, :top_k]
    desc = F.normalize(torch.stack([O.sample_bicubic(fn[b], mk[b], H, W) for b in range(B)]), dim=-1)
    valid = scores > 0
    return [{"keypoints": mk[b][valid[b]].float(), "scores": scores[b][valid[b]], "descriptors": desc[b][valid[b]]} for b in range(B)]


def run_rel_head(bins, sd, feats, fx):
    B, _, h, w = feats.shape
    cl = feats.permute(0, 2, 3, 1).reshape(-1, 64).contiguous()
    ws, bs = zip(*[(w_.view(64, 64), b_) for w_, b_ in (fold(sd, f"heatmap_head.{i}") for i in range(2))])
    blob = np.concatenate([np.array([0, fx, len(cl), 0, 0], np.int32).view(np.float32)] + [np.asarray(a, np.float32).reshape(-1) for a in
                          [cl] + list(ws) + [sd["heatmap_head.2.weight"].view(64).float()] + list(bs) + [sd["heatmap_head.2.bias"].float()]]).tobytes()
    out = subprocess.run([bins[1]], input=blob, capture_output=True, check=True, timeout=3000).stdout
    assert int(np.frombuffer(out[-4:], np.int32)[0]) == 0
    return torch.from_numpy(np.frombuffer(out[:4 * len(cl)], np.float32).reshape(B, 1, h, w).copy())


def conv_emu(bins, kind, hdr, x, tensors, shape, cl=False, status=True):
    blob = np.concatenate([np.array(hdr, np.int32).view(np.float32)] + [np.asarray(a, np.float32).reshape(-1) for a in [x] + tensors]).tobytes()
    out = subprocess.run([bins.by_name[kind]], input=blob, capture_output=True, check=True, timeout=3000).stdout
    if status:
        assert int(np.frombuffer(out[-4:], np.int32)[0]) == 0, (kind, hdr, "an activation left the range of the fp16 pair")
        out = out[:-4]
    y = torch.from_numpy(np.frombuffer(out, np.float32).copy())
    return y.view(shape[0], shape[2], shape[3], shape[1]).permute(0, 3, 1, 2).contiguous() if cl else y.view(shape)


def rest_of_backbone_emulated(bins, sd, x1):
    """block2 .. block_fusion.2 on the kernels the backbone routes the bench batch to by default: the 24-channel
    layers on conv_bx_kernel / conv_bxs2_kernel, every 64 -> 64 3x3 layer on conv_rs64_kernel (weights resident in registers; block3.1 + 3.2 and block_fusion.1 + .2
    with the trailing 1x1 fused, the latter channels-last), block5.1 / 5.2 on its 128-channel form, block4.0 / block5.0 on conv_bx64s2x_kernel -- block5.3 inside pyramid53_kernel
    (the pyramid sum with the 1x1 fused in): all 17 convolution layers.  Every kernel's range flag must stay clear on the fixtures."""
    B, _, H4, W4 = x1.shape
    H8, W8, H16, W16, H32, W32 = H4 // 2, W4 // 2, H4 // 4, W4 // 4, H4 // 8, W4 // 8
    a = conv_emu(bins, "conv_bx24_emu", [B, H4, W4, 1, 1, 5], x1, list(fold(sd, "block2.0")), (B, 24, H4, W4))
    a = conv_emu(bins, "conv_bx24_emu", [B, H4, W4, 1, 1, 5], a, list(fold(sd, "block2.1")), (B, 24, H4, W4))
    x3 = conv_emu(bins, "conv_bx24_emu", [B, H4, W4, 2, 1, 5], a, list(fold(sd, "block3.0")), (B, 64, H8, W8))
    w2, b2 = fold(sd, "block3.2")
    x3 = conv_emu(bins, "conv_rs64_emu", [B, H8, W8, 1, 3, 0, 1, 1], x3, list(fold(sd, "block3.1")) + [w2.view(64, 64), b2], (B, 64, H8, W8))
    x4 = conv_emu(bins, "conv_bx64s2_emu", [B, H8, W8, 64, 1, 3], x3, list(fold(sd, "block4.0")), (B, 64, H16, W16))
    x4 = conv_emu(bins, "conv_rs64_emu", [B, H16, W16, 1, 3, 0, 0, 0], x4, list(fold(sd, "block4.1")), (B, 64, H16, W16))
    x4 = conv_emu(bins, "conv_rs64_emu", [B, H16, W16, 1, 2, 0, 0, 0], x4, list(fold(sd, "block4.2")), (B, 64, H16, W16))
    x5 = conv_emu(bins, "conv_bx64s2_emu", [B, H16, W16, 128, 1, 2], x4, list(fold(sd, "block5.0")), (B, 128, H32, W32))
    x5 = conv_emu(bins, "conv_rs64_emu", [B, H32, W32, 1, 2, 0, 128, 0], x5, list(fold(sd, "block5.1")), (B, 128, H32, W32))
    x5 = conv_emu(bins, "conv_rs64_emu", [B, H32, W32, 1, 1, 0, 128, 0], x5, list(fold(sd, "block5.2")), (B, 128, H32, W32))
    w53, b53 = fold(sd, "block5.3")
    f = conv_emu(bins, "pyramid53_emu", [B, H8, W8, H16, W16, H32, W32, 1], x3, [x4, x5, w53.view(64, 128).t().contiguous(), b53], (B, 64, H8, W8), status=False)      # pyramid53_kernel (sliced, shipped): block5.3 + the pyramid sum
    f = conv_emu(bins, "conv_rs64_emu", [B, H8, W8, 1, 3, 0, 0, 0], f, list(fold(sd, "block_fusion.0")), (B, 64, H8, W8))
    return conv_emu(bins, "conv_rs64_emu", [B, H8, W8, 1, 3, 0, 2, 0], f, list(fold(sd, "block_fusion.1")) + [sd["block_fusion.2.weight"].view(64, 64).float(), sd["block_fusion.2.bias"].float()],
                    (B, 64, H8, W8), cl=True)


@pytest.mark.parametrize("which", ["g1_small", "g2_vga_pair"])
def test_default_block1_and_head_keep_the_references_key_points(bins, which):
    sd = fixtures.synthetic_state_dict(0)
    with torch.inference_mode():
        if which == "g1_small":
            g = np.load(os.path.join(ROOT, "tests", "golden", "g1_small.npz")); x = fixtures.texture_images(2, 96, 128, seed=11); top_k = 256
            gold = [{k: g[f"{k}{b}"] for k in ("keypoints", "scores", "descriptors")} for b in range(2)]
        else:
            g = np.load(os.path.join(ROOT, "tests", "golden", "g2_vga_pair.npz")); a, b_ = fixtures.shifted_pair(1, 480, 640, seed=7); x = torch.cat([a, b_]); top_k = 4096
            gold = [{"keypoints": g[f"kp_{t}"].astype(np.float32), "scores": g[f"sc_{t}"]} for t in ("a", "b")]
        B, _, H, W = x.shape
        gray, coef = gray_coef(x)
        _, _, _, taps = O.backbone(sd, x, keep=True)
        oheat = O.kpts_heatmap(taps["logits"])
        errs = {}
        for tag, mode, fx in (("fallback", 5, -1), ("default", 7, 1)):
            x1 = run_block1(bins, sd, gray, coef, mode)
            heat = run_head(bins, sd, gray, coef, fx)
            feats, rel = rest_of_backbone(sd, x1)
            errs[tag] = e = {"x1": float((x1 - taps["x1"]).abs().max()), "feats": float((feats - taps["feats"]).abs().max()),
                             "rel": float((rel - taps["reliability"]).abs().max()), "heat": float((heat - oheat).abs().max())}
            assert e["x1"] <= 2e-5 and e["feats"] <= 1e-4 and e["rel"] <= 3e-5 and e["heat"] <= 1e-5, (tag, e)      # the GPU suite's tolerances against the oracle
            out = detect(feats, heat, rel, top_k, H, W)
            for b in range(B):
                gd, t = dict(gold[b]), dict(out[b])
                if "descriptors" not in gd:      # (g2 holds every 8th descriptor row only: the lists are compared)
                    gd["descriptors"] = np.zeros((len(gd["keypoints"]), 64), np.float32); t["descriptors"] = torch.zeros(len(t["keypoints"]), 64)
                rep = parity.compare_keypoints(t, gd, heat=oheat[b, 0])      # raises on anything that is not a tie in the reference's own maps
                print(which, tag, "image", b, rep)
                assert rep["common"] == rep["n_ref"] == top_k and rep["exceptions"] == 0, (tag, rep)      # the SAME key-point set as the reference, both forms
                assert rep.get("rank_moved", 0) <= 64 and rep.get("rank_moved_maxgap", 0.0) <= 5e-6, (tag, rep)      # (rank moves: only among scores a few ulps apart)
        print(which, errs)
        for k in ("x1", "feats", "rel", "heat"):
            assert errs["default"][k] <= 1.5 * errs["fallback"][k] + 1e-6, (k, errs)      # no further from the reference than the fp32-range forms


@pytest.mark.parametrize("which", ["g1_small", pytest.param("g2_vga_pair", marks=pytest.mark.skipif(not os.environ.get("XFH_EMU_VGA"), reason="minutes of emulation: XFH_EMU_VGA=1"))])
def test_every_default_kernel_in_one_chain_keeps_the_references_key_points(bins, which):
    """The whole default path at once: block1 mode 7, 16 of the 17 convolution layers on the kernels of rest_of_backbone_emulated, both heads in the fp16-pair form --
    the reference's key-point sets, no range flag on the fixtures' real weights and activations (a flag would send the model to the fallback and make the forms pointless)."""
    sd = fixtures.synthetic_state_dict(0)
    with torch.inference_mode():
        if which == "g1_small":
            g = np.load(os.path.join(ROOT, "tests", "golden", "g1_small.npz")); x = fixtures.texture_images(2, 96, 128, seed=11); top_k = 256
            gold = [{k: g[f"{k}{b}"] for k in ("keypoints", "scores", "descriptors")} for b in range(2)]
        else:
            g = np.load(os.path.join(ROOT, "tests", "golden", "g2_vga_pair.npz")); x = torch.cat(fixtures.shifted_pair(1, 480, 640, seed=7)); top_k = 4096
            gold = [{"keypoints": g[f"kp_{t}"].astype(np.float32), "scores": g[f"sc_{t}"]} for t in ("a", "b")]
        B, Cc, H, W = x.shape
        o_ = subprocess.run([bins.by_name["gray_emu"]], input=np.concatenate([np.array([B, Cc, H, W], np.int32).view(np.float32), x.numpy().reshape(-1)]).tobytes(),
                            capture_output=True, check=True, timeout=600).stdout      # gray_stats_kernel + gray_coef_kernel (sliced, shipped)
        gray = torch.from_numpy(np.frombuffer(o_[:4 * B * H * W], np.float32).reshape(B, H, W).copy())
        coef = torch.from_numpy(np.frombuffer(o_[4 * B * H * W:], np.float32).reshape(B, 2).copy())
        _, _, _, taps = O.backbone(sd, x, keep=True)
        oheat = O.kpts_heatmap(taps["logits"])
        x1 = run_block1(bins, sd, gray, coef, 7)
        feats = rest_of_backbone_emulated(bins, sd, x1)
        rel, heat = run_rel_head(bins, sd, feats, 1), run_head(bins, sd, gray, coef, 1)
        e = {"x1": float((x1 - taps["x1"]).abs().max()), "feats": float((feats - taps["feats"]).abs().max()),
             "rel": float((rel - taps["reliability"]).abs().max()), "heat": float((heat - oheat).abs().max())}
        print(which, "every default kernel", e)
        assert e["x1"] <= 2e-5 and e["feats"] <= 1e-4 and e["rel"] <= 3e-5 and e["heat"] <= 1e-5, e
        for b, out in enumerate(detect(feats, heat, rel, top_k, H, W)):
            gd, t = dict(gold[b]), dict(out)
            if "descriptors" not in gd:
                gd["descriptors"] = np.zeros((len(gd["keypoints"]), 64), np.float32); t["descriptors"] = torch.zeros(len(t["keypoints"]), 64)
            rep = parity.compare_keypoints(t, gd, heat=oheat[b, 0])
            print(which, "image", b, rep)
            assert rep["common"] == rep["n_ref"] == top_k and rep["exceptions"] == 0, rep
            assert rep.get("rank_moved", 0) <= 64 and rep.get("rank_moved_maxgap", 0.0) <= 5e-6, rep

