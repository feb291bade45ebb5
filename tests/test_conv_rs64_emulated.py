"""conv_rs64_kernel's body (csrc/conv_rs64_body.hpp: 64 -> 64 3x3 convolution in the fp16-pair arithmetic with the weights resident in registers, K split over the four waves
of a workgroup, padded-raster walk of the map, partial sums reduced through LDS) compiled for the HOST (tests/emu/) against a float64 convolution."""
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
    out = os.path.join(tempfile.mkdtemp(), "conv_rs64_emu")
    subprocess.run([CLANG, "-O1", "-w", "-std=c++20", "-pthread", "-I", os.path.join(ROOT, "accelerated_features_amd", "csrc"), "-I", os.path.join(ROOT, "tests", "emu"),
                    os.path.join(ROOT, "tests", "emu", "conv_rs64_emu.cpp"), "-o", out], check=True)
    return out


def run(emu_bin, x, w, b, relu, grid, k, fuse=0, w2=None, b2=None, relu2=0):
    B, _, H, W = x.shape
    arrs = [x, w, b] + ([w2, b2] if fuse else [])
    blob = np.concatenate([np.array([B, H, W, relu, grid, k, fuse, relu2], np.int32).view(np.float32)] + [t.numpy().astype(np.float32).reshape(-1) for t in arrs])
    out = subprocess.run([emu_bin], input=blob.tobytes(), capture_output=True, check=True, timeout=240).stdout
    y = np.frombuffer(out[:-4], np.float32)
    y = y.reshape(B, H, W, 64).transpose(0, 3, 1, 2) if fuse == 2 else y.reshape(B, 64, H, W)
    return y, int(np.frombuffer(out[-4:], np.int32)[0])


# (30 x 40: the 1/16-scale VGA map, nseg 4, runs of 5 units; 15 x 80: the 1/8-scale pitch, nseg 5; 7 x 33: a map smaller than a ring; 9 x 93: nseg 4 at its widest;
#  6 x 125: the widest map ONE ring holds (nseg 5);  k = 1: whole images per run, two runs for one workgroup; k = nu: one unit per run;
#  5 x 150, 4 x 260: wider maps run as two / three column strips, the last one narrower (260 = 87 + 87 + 86), pad columns = the neighbouring strips' pixels)
@pytest.mark.parametrize("shape,relu,grid,k", [((1, 30, 40), 1, 4, 0), ((2, 15, 80), 0, 3, 2), ((3, 7, 33), 1, 2, 1), ((1, 9, 93), 1, 2, 0), ((1, 6, 125), 0, 2, 0), ((1, 12, 20), 1, 64, 0), ((1, 1, 1), 0, 1, 0),
                                               ((2, 5, 150), 1, 3, 0), ((1, 4, 260), 0, 4, 2), ((1, 3, 126), 1, 2, 1)])
def test_conv_rs64_body_on_the_host(emu_bin, shape, relu, grid, k):
    B, H, W = shape
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn(B, 64, H, W, generator=g) * 2              # (signed inputs: block_fusion.0 reads the pyramid sum)
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    b = torch.randn(64, generator=g) * 0.3
    y, status = run(emu_bin, x, w, b, relu, grid, k)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    if relu:
        ref = torch.relu(ref)
    d = np.abs(y - ref.numpy())
    print(f"{shape} relu {relu} grid {grid} k {k}: max |err| {d.max():.3g}, max |y| {float(ref.abs().max()):.3g}")
    assert status == 0 and np.isfinite(y).all()
    assert d.max() <= 3e-6 * float(ref.abs().max())


def test_conv_rs64_reports_its_range(emu_bin):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 64, 6, 10, generator=g)
    x[0, 37, 3, 4] = 7e4
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    _, status = run(emu_bin, x, w, torch.zeros(64), 1, 2, 0)
    assert status == 1


# the trailing 1x1 (block3.2 behind block3.1: NCHW; block_fusion.2 behind block_fusion.1: channels-last) fused: a block's 3x3 outputs go through LDS as fp16 pairs, its 1x1 runs
# inside the MFMAs of the block after next
@pytest.mark.parametrize("fuse,shape,relu2,grid,k", [(1, (1, 30, 40), 0, 4, 0), (2, (2, 15, 80), 0, 3, 2), (2, (3, 7, 33), 1, 2, 1), (1, (1, 9, 93), 0, 2, 0), (2, (1, 1, 1), 0, 1, 0), (1, (1, 5, 7), 1, 8, 0),
                                                     (2, (1, 5, 128), 0, 3, 0), (1, (2, 3, 94), 1, 2, 1)])      # (128, 94 columns: two strips under the fused form's 93-column rings)
def test_conv_rs64_with_the_trailing_1x1_fused(emu_bin, fuse, shape, relu2, grid, k):
    B, H, W = shape
    g = torch.Generator().manual_seed(H * W + fuse)
    x = torch.randn(B, 64, H, W, generator=g) * 2
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    b = torch.randn(64, generator=g) * 0.3
    w2 = torch.randn(64, 64, generator=g) / 8
    b2 = torch.randn(64, generator=g) * 0.3
    y, status = run(emu_bin, x, w, b, 1, grid, k, fuse, w2, b2, relu2)
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1))
    ref = torch.nn.functional.conv2d(ref, w2.double().view(64, 64, 1, 1), b2.double())
    if relu2:
        ref = torch.relu(ref)
    d = np.abs(y - ref.numpy())
    print(f"fuse {fuse} {shape} relu2 {relu2} grid {grid} k {k}: max |err| {d.max():.3g}, max |y| {float(ref.abs().max()):.3g}")
    assert status == 0 and np.isfinite(y).all()
    assert d.max() <= 3e-6 * float(ref.abs().max())


# the 128 -> 128 layers (block5.1, block5.2) on the same body: a workgroup computes a quarter of the couts, a wave multiplies 32 input channels (two chunks, one accumulator each)
@pytest.mark.parametrize("shape,relu,groups,k", [((2, 15, 20), 1, 2, 0), ((1, 7, 33), 0, 1, 2), ((3, 4, 5), 1, 2, 1), ((1, 3, 61), 1, 1, 0), ((1, 3, 100), 1, 2, 0)])      # (100 columns: two strips)
def test_conv_rs64_body_with_128_channels(emu_bin, shape, relu, groups, k):
    B, H, W = shape
    g = torch.Generator().manual_seed(H * W + 128)
    x = torch.randn(B, 128, H, W, generator=g) * 2
    w = torch.randn(128, 128, 3, 3, generator=g) / 34
    b = torch.randn(128, generator=g) * 0.3
    blob = np.concatenate([np.array([B, H, W, relu, groups, k, 128, 0], np.int32).view(np.float32)] + [t.numpy().astype(np.float32).reshape(-1) for t in (x, w, b)])
    out = subprocess.run([emu_bin], input=blob.tobytes(), capture_output=True, check=True, timeout=400).stdout
    y = np.frombuffer(out[:-4], np.float32).reshape(B, 128, H, W)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1)
    if relu:
        ref = torch.relu(ref)
    d = np.abs(y - ref.numpy())
    print(f"128 channels {shape} relu {relu} groups {groups} k {k}: max |err| {d.max():.3g}, max |y| {float(ref.abs().max()):.3g}")
    assert int(np.frombuffer(out[-4:], np.int32)[0]) == 0 and np.isfinite(y).all()
    assert d.max() <= 3e-6 * float(ref.abs().max())


def test_row_of_a_padded_raster_position_is_exact():
    """conv_rs64_body.hpp: row_of(i) = (int)((i + 0.5f) * (1.f / P)) replaces the integer division of a position by the raster's pitch: exact for every position below 2^20 and
    every pitch the kernel's rings admit (the launcher refuses larger maps)"""
    i = np.arange(0, 1 << 20, dtype=np.int64)
    for P in range(3, 128):
        r = ((i.astype(np.float32) + np.float32(0.5)) * (np.float32(1.0) / np.float32(P))).astype(np.int32)
        assert (r == i // P).all(), P
