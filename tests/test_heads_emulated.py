"""The head kernels (csrc/head_bx_body.hpp: the default, fp16-pair arithmetic -- three MFMAs per product, accumulators at scale 2^11, bias and scale applied by the consumer,
the weight image of weight_split.hpp: pack_head_layer; csrc/head_f32r_body.hpp: the f32-MFMA fallback) compiled for the HOST (tests/emu/) against a float64 reference."""
import os
import subprocess
import tempfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def emu_bin():
    if not os.path.exists(CLANG):
        pytest.skip("no host clang")
    out = os.path.join(tempfile.mkdtemp(), "head_emu")
    subprocess.run([CLANG, "-O1", "-w", "-std=c++20", "-pthread", "-I", os.path.join(ROOT, "accelerated_features_amd", "csrc"), "-I", os.path.join(ROOT, "tests", "emu"),
                    os.path.join(ROOT, "tests", "emu", "head_emu.cpp"), "-o", out], check=True)
    return out


def _blob(hdr, arrs):
    return np.concatenate([np.array(hdr, np.int32).view(np.float32)] + [np.asarray(a, np.float32).reshape(-1) for a in arrs]).tobytes()


@pytest.mark.parametrize("fx", [-1, 1])      # 1: the fp16-pair heads (the default), -1: the f32-MFMA heads with register input (the range fallback)
def test_keypoint_head_on_the_host(emu_bin, fx):
    g = torch.Generator().manual_seed(4 + fx)
    B, H, W = 2, 96, 136                              # 2 x 12 x 17 = 408 cells: one full tile and a partial one
    gray = torch.rand(B, H, W, generator=g)
    coef = torch.stack([1.0 + torch.rand(B, generator=g) * 2, torch.randn(B, generator=g) * 0.5], 1)
    ws = [torch.randn(64, 64, generator=g) * 0.18 for _ in range(3)] + [torch.randn(65, 64, generator=g) * 0.3]
    bs = [torch.randn(64, generator=g) * 0.3 for _ in range(3)] + [torch.randn(65, generator=g)]
    out = subprocess.run([emu_bin], input=_blob([1, fx, B, H, W], [gray, coef] + ws + bs), capture_output=True, check=True, timeout=240).stdout
    ncell = B * (H // 8) * (W // 8)
    heat = np.frombuffer(out[:4 * B * H * W], np.float32).reshape(B, H, W)
    logits = np.frombuffer(out[4 * B * H * W:4 * B * H * W + 4 * ncell * 65], np.float32).reshape(ncell, 65)
    status = int(np.frombuffer(out[-4:], np.int32)[0])
    # reference: 8 x 8 unfold (channel = 8 dy + dx) -> 3 x (linear + ReLU) -> linear -> softmax, dustbin dropped, depth-to-space
    x = (gray.double() * coef[:, 0].double().view(-1, 1, 1) + coef[:, 1].double().view(-1, 1, 1))
    u = x.view(B, H // 8, 8, W // 8, 8).permute(0, 1, 3, 2, 4).reshape(ncell, 64)
    a = u
    for w, b in zip(ws[:3], bs[:3]):
        a = torch.relu(a @ w.double().T + b.double())
    lg = a @ ws[3].double().T + bs[3].double()
    sm = torch.softmax(lg, 1)[:, :64]
    href = sm.view(B, H // 8, W // 8, 8, 8).permute(0, 1, 3, 2, 4).reshape(B, H, W)
    e_l, e_h = float(np.abs(logits - lg.numpy()).max()), float(np.abs(heat - href.numpy()).max())
    print(f"fx {fx}: logits max |err| {e_l:.3g} (max |logit| {float(lg.abs().max()):.3g}), heat max |err| {e_h:.3g}")
    assert status == 0 and np.isfinite(heat).all()
    assert e_l <= 2e-5 * float(lg.abs().max()) and e_h <= 1e-6


@pytest.mark.parametrize("fx", [-1, 1])
def test_keypoint_head_on_the_host_reproduces_the_reference_made_golden(emu_bin, fx):
    """The same kernel bodies on the fixture of tests/golden/g1_small.npz (image, synthetic weights with the calibrated BatchNorm statistics; heat map and logits written by the
    UNMODIFIED reference, tests/golden/make_golden.py): every form inside the tolerances the GPU suite applies to that golden (heat 1e-5, logits 5e-4) -- with margin, and
    before any of the prepared forms has met the suite on a GPU.  BatchNorm (affine=False, eps 1e-5) folded here the way xfh_create folds it."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fixtures
    sd = fixtures.synthetic_state_dict(0)
    g = np.load(os.path.join(ROOT, "tests", "golden", "g1_small.npz"))
    x = fixtures.texture_images(2, 96, 128, seed=11)
    B, _, H, W = x.shape
    gray = x.mean(1)
    gd = gray.double()
    alpha = 1.0 / torch.sqrt(gd.var((1, 2), unbiased=False) + 1e-5)               # InstanceNorm2d(1): x * alpha + beta  (modules/model.py:151)
    coef = torch.stack([alpha, -gd.mean((1, 2)) * alpha], 1).float()
    ws, bs = [], []
    for i in range(3):
        s_ = 1.0 / torch.sqrt(sd[f"keypoint_head.{i}.layer.1.running_var"].double() + 1e-5)
        ws.append((sd[f"keypoint_head.{i}.layer.0.weight"].double().view(64, 64) * s_[:, None]).float())
        bs.append((-sd[f"keypoint_head.{i}.layer.1.running_mean"].double() * s_).float())
    ws.append(sd["keypoint_head.3.weight"].view(65, 64).float()); bs.append(sd["keypoint_head.3.bias"].float())
    out = subprocess.run([emu_bin], input=_blob([1, fx, B, H, W], [gray, coef] + ws + bs), capture_output=True, check=True, timeout=240).stdout
    ncell = B * (H // 8) * (W // 8)
    heat = np.frombuffer(out[:4 * B * H * W], np.float32).reshape(B, 1, H, W)
    logits = np.frombuffer(out[4 * B * H * W:4 * B * H * W + 4 * ncell * 65], np.float32).reshape(ncell, 65)
    gl = torch.from_numpy(g["logits"]).permute(0, 2, 3, 1).reshape(ncell, 65).numpy()
    e_h, e_l = float(np.abs(heat - g["heat"]).max()), float(np.abs(logits - gl).max())
    print(f"fx {fx}: heat max |err| vs the reference's {e_h:.3g}, logits {e_l:.3g} (max |logit| {float(np.abs(gl).max()):.3g})")
    assert int(np.frombuffer(out[-4:], np.int32)[0]) == 0
    assert e_h <= 5e-6 and e_l <= 1e-4                    # (GPU suite: 1e-5 / 5e-4; measured here: 3.1e-6 .. 3.9e-6 / 2.3e-5 .. 2.5e-5)


@pytest.mark.parametrize("fx", [-1, 1])
def test_reliability_head_on_the_host(emu_bin, fx):
    g = torch.Generator().manual_seed(14 + fx)
    n = 300
    feats = torch.randn(n, 64, generator=g) * 2
    ws = [torch.randn(64, 64, generator=g) * 0.18 for _ in range(2)]
    w2 = torch.randn(64, generator=g) * 0.2
    bs = [torch.randn(64, generator=g) * 0.3 for _ in range(2)]
    b2 = torch.randn(1, generator=g)
    out = subprocess.run([emu_bin], input=_blob([0, fx, n, 0, 0], [feats] + ws + [w2] + bs + [b2]), capture_output=True, check=True, timeout=240).stdout
    rel = np.frombuffer(out[:4 * n], np.float32)
    inv = np.frombuffer(out[4 * n:8 * n], np.float32)
    status = int(np.frombuffer(out[-4:], np.int32)[0])
    a = feats.double()
    for w, b in zip(ws, bs):
        a = torch.relu(a @ w.double().T + b.double())
    ref = torch.sigmoid(a @ w2.double() + b2.double())
    iref = 1.0 / feats.double().norm(dim=1).clamp_min(1e-12)
    e_r, e_i = float(np.abs(rel - ref.numpy()).max()), float(np.abs(inv / iref.numpy() - 1).max())
    print(f"fx {fx}: reliability max |err| {e_r:.3g}, 1 / |feats| max rel err {e_i:.3g}")
    assert status == 0 and e_r <= 2e-6 and e_i <= 1e-6
    if fx > 0:      # the range guard: features beyond the fp16 range are reported
        out = subprocess.run([emu_bin], input=_blob([0, fx, n, 0, 0], [feats * 1.0e5] + ws + [w2] + bs + [b2]), capture_output=True, check=True, timeout=240).stdout
        assert int(np.frombuffer(out[-4:], np.int32)[0]) & 1
