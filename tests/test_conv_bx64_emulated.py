"""conv_bx64_kernel's body (csrc/conv_bx64_body.hpp: the 64 -> 64 3x3 convolutions on split-operand MFMAs, alone and with their trailing 1x1 fused, NCHW or
channels-last output) compiled for the HOST (tests/emu/) against a float64 convolution: the bf16 three-way split and the fp16-pair form, both GPU-validated -- here
as the emulator's control for this kernel family, so that what round 5 changes in it can be checked before it meets the hardware."""
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
    out = os.path.join(tempfile.mkdtemp(), "conv_bx64_emu")
    subprocess.run([CLANG, "-O1", "-w", "-std=c++20", "-pthread", "-I", os.path.join(ROOT, "accelerated_features_amd", "csrc"), "-I", os.path.join(ROOT, "tests", "emu"),
                    os.path.join(ROOT, "tests", "emu", "conv_bx64_emu.cpp"), "-o", out], check=True)
    return out


@pytest.mark.parametrize("fuse,fx,shape,grid", [(0, 1, (1, 24, 40), 3), (0, 0, (1, 16, 16), 2), (1, 1, (2, 24, 32), 5), (2, 1, (1, 18, 20), 2), (0, 1, (8, 8, 16), 8), (0, 2, (1, 24, 40), 3), (1, 2, (1, 16, 32), 2), (2, 2, (1, 18, 20), 2)])
def test_conv_bx64_body_on_the_host(emu_bin, fuse, fx, shape, grid):
    B, H, W = shape                                   # (24 x 40: full tiles, a half tile and a partial strip; 18 x 20: rows and columns beyond the map; B = 8, grid 8: the XCD mapping)
    g = torch.Generator().manual_seed(10 * fuse + fx)
    x = torch.relu(torch.randn(B, 64, H, W, generator=g)) * 2
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    b = torch.randn(64, generator=g) * 0.3
    w2 = torch.randn(64, 64, generator=g) / 8
    b2 = torch.randn(64, generator=g) * 0.3
    arrs = [x, w, b] + ([w2, b2] if fuse else [])
    blob = np.concatenate([np.array([B, H, W, fuse, fx, 1, 0, grid], np.int32).view(np.float32)] + [t.numpy().astype(np.float32).reshape(-1) for t in arrs])
    out = subprocess.run([emu_bin], input=blob.tobytes(), capture_output=True, check=True, timeout=240).stdout
    y = np.frombuffer(out[:-4], np.float32)
    status = int(np.frombuffer(out[-4:], np.int32)[0])
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1))
    if fuse:
        ref = torch.nn.functional.conv2d(ref, w2.double().view(64, 64, 1, 1), b2.double())
    y = y.reshape(B, H, W, 64).transpose(0, 3, 1, 2) if fuse == 2 else y.reshape(B, 64, H, W)
    d = np.abs(y - ref.numpy())
    print(f"fuse {fuse} fx {fx} {shape}: max |err| {d.max():.3g}, max |y| {float(ref.abs().max()):.3g}")
    assert status == 0 and np.isfinite(y).all()
    assert d.max() <= 3e-6 * float(ref.abs().max())


@pytest.mark.parametrize("shape,grid", [((1, 24, 40), 3), ((2, 18, 20), 4), ((8, 8, 16), 8)])
def test_split_format_link_on_the_host(emu_bin, shape, grid):
    """producer-side split: a plain 3x3 writes its output as fp16 pairs in the record format of conv_bx64_body.hpp, the next layer (3x3 + 1x1, channels-last) stages it by
    LDS-DMA alone (swizzled 64-byte records, double-buffered chunks, zeros for the halo outside the map): the pair against two float64 convolutions"""
    B, H, W = shape
    g = torch.Generator().manual_seed(H)
    x = torch.relu(torch.randn(B, 64, H, W, generator=g)) * 2
    wA, wB = (torch.randn(64, 64, 3, 3, generator=g) / 24 for _ in range(2))
    bA, bB, b2 = (torch.randn(64, generator=g) * 0.3 for _ in range(3))
    w2 = torch.randn(64, 64, generator=g) / 8
    blob = np.concatenate([np.array([B, H, W, 3, 1, 1, 0, grid], np.int32).view(np.float32)] + [t.numpy().astype(np.float32).reshape(-1) for t in (x, wA, bA, wB, bB, w2, b2)])
    out = subprocess.run([emu_bin], input=blob.tobytes(), capture_output=True, check=True, timeout=240).stdout
    y = np.frombuffer(out[:-4], np.float32).reshape(B, H, W, 64).transpose(0, 3, 1, 2)
    F = torch.nn.functional
    ref = torch.relu(F.conv2d(x.double(), wA.double(), bA.double(), padding=1))
    ref = torch.relu(F.conv2d(ref, wB.double(), bB.double(), padding=1))
    ref = F.conv2d(ref, w2.double().view(64, 64, 1, 1), b2.double())
    d = np.abs(y - ref.numpy())
    print(f"split-format link {shape}: max |err| {d.max():.3g}, max |y| {float(ref.abs().max()):.3g}")
    assert int(np.frombuffer(out[-4:], np.int32)[0]) == 0 and np.isfinite(y).all() and d.max() <= 5e-6 * float(ref.abs().max())
