"""The fused block1 + skip1 kernel body (csrc/block1_body.hpp) compiled for the HOST (tests/emu/: one host thread per work-item, LDS as a buffer, the LDS-DMA
and v_mfma_f32_16x16x32_f16 emulated) and run against a float64 convolution reference: the shipped vector form (mode 5) and the matrix-core forms (6: block1.3,
7: block1.2 + block1.3 in the fp16-pair arithmetic) on images whose last tiles are partial.  What the numpy model of test_block1_fx_model.py says about the
layouts, this says about the kernel source itself -- without a GPU."""
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
    td = tempfile.mkdtemp()
    out = os.path.join(td, "block1_emu")
    subprocess.run([CLANG, "-O1", "-w", "-std=c++20", "-pthread", "-I", os.path.join(ROOT, "accelerated_features_amd", "csrc"), "-I", os.path.join(ROOT, "tests", "emu"),
                    os.path.join(ROOT, "tests", "emu", "block1_emu.cpp"), "-o", out], check=True)
    return out


def _case(seed, B, H, W):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).float()
    w = {"w1": r(4, 1, 3, 3, k=0.5), "b1": r(4, k=0.2), "w2": r(8, 4, 3, 3, k=0.25), "b2": r(8, k=0.2), "w3": r(8, 8, 3, 3, k=0.2), "b3": r(8, k=0.2),
         "w4": r(24, 8, 3, 3, k=0.2), "b4": r(24, k=0.2), "skw": r(24, 1, 1, 1, k=0.5), "skb": r(24, k=0.2)}
    gray = torch.rand(B, 1, H, W, generator=g)
    coef = torch.stack([0.5 + torch.rand(B, generator=g) * 3, torch.randn(B, generator=g)], 1).float()      # per-image {alpha, beta}
    return w, gray.float(), coef


def _reference(w, gray, coef):
    F = torch.nn.functional
    d = lambda t: t.double()
    x = d(gray) * d(coef[:, 0]).view(-1, 1, 1, 1) + d(coef[:, 1]).view(-1, 1, 1, 1)
    a = F.relu(F.conv2d(x, d(w["w1"]), d(w["b1"]), padding=1))
    a = F.relu(F.conv2d(a, d(w["w2"]), d(w["b2"]), stride=2, padding=1))
    a = F.relu(F.conv2d(a, d(w["w3"]), d(w["b3"]), padding=1))
    a = F.relu(F.conv2d(a, d(w["w4"]), d(w["b4"]), stride=2, padding=1))
    return a + F.conv2d(F.avg_pool2d(x, 4, 4), d(w["skw"]), d(w["skb"]))


def _run(emu_bin, mode, w, gray, coef):
    B, _, H, W = gray.shape
    kc = lambda t: t.permute(1, 2, 3, 0).reshape(-1).contiguous()          # (cout, cin, 3, 3) -> [(ci * 9 + tap) * cout + co]
    pad = lambda t: torch.cat([t.reshape(-1), torch.zeros(32 - t.numel())])
    blob = np.concatenate([np.array([B, H, W, mode], np.int32).view(np.float32)] + [t.numpy().astype(np.float32).reshape(-1) for t in (
        gray, coef, kc(w["w1"]), w["b1"], kc(w["w2"]), w["b2"], kc(w["w3"]), w["b3"], kc(w["w4"]), pad(w["b4"]), pad(w["skw"]), pad(w["skb"]))])
    out = subprocess.run([emu_bin], input=blob.tobytes(), capture_output=True, check=True, timeout=240).stdout
    x1 = np.frombuffer(out[:-4], np.float32).reshape(B, 24, H // 4, W // 4)
    return x1, int(np.frombuffer(out[-4:], np.int32)[0])


@pytest.mark.parametrize("mode", [5, 7])
def test_block1_body_on_the_host_is_the_network(emu_bin, mode):
    for seed, (B, H, W) in enumerate(((1, 64, 64), (2, 96, 160))):        # 2 x 1 full tiles; 3 x 2.5 tiles per image (a partial last column of tiles)
        w, gray, coef = _case(seed, B, H, W)
        ref = _reference(w, gray, coef).numpy()
        x1, status = _run(emu_bin, mode, w, gray, coef)
        assert np.isfinite(x1).all() and status == 0, (mode, status)
        d = np.abs(x1 - ref)
        i = np.unravel_index(int(d.argmax()), d.shape)
        print(f"mode {mode} ({B},{H},{W}): max |err| {d.max():.3g} at {i}, max |x1| {np.abs(ref).max():.3g}")
        assert d.max() <= 2e-5 * max(1.0, float(np.abs(ref).max())), (mode, (B, H, W), float(d.max()), i)


def test_block1_body_reports_the_range_of_the_pair(emu_bin):
    w, gray, coef = _case(9, 1, 32, 32)
    coef = coef * 0 + torch.tensor([[3.0e6, 0.0]])
    assert _run(emu_bin, 5, w, gray, coef)[1] == 0
    assert _run(emu_bin, 7, w, gray, coef)[1] & 1
