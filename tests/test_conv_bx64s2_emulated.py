"""conv_bx64s2x_kernel's body (csrc/conv_bx64s2_body.hpp: the stride-2 64 -> 64 | 128 convolutions, block4.0 / block5.0, in the fp16-pair arithmetic with the split of the next
chunk done by four staging waves beside eight multiplying ones) compiled for the HOST (tests/emu/) against a float64 convolution."""
import os
import subprocess
import tempfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


# The weight ring's LDS-DMA is emulated twice: copied at issue (the earliest the hardware could deliver it) and copied as late as the kernel's own s_waitcnt vmcnt(n) allow
# (EMU_DEFER_DMA: a thread's requests queue up, a wait completes all but the n youngest) -- a three-slot ring requested two rows ahead must be right under both.
@pytest.fixture(scope="module", params=["dma_at_issue", "dma_at_wait"])
def emu_bin(request):
    if not os.path.exists(CLANG):
        pytest.skip("no host clang")
    out = os.path.join(tempfile.mkdtemp(), "conv_bx64s2_emu")
    subprocess.run([CLANG, "-O1", "-w", "-std=c++20", "-pthread"] + (["-DEMU_DEFER_DMA"] if request.param == "dma_at_wait" else []) + ["-I", os.path.join(ROOT, "accelerated_features_amd", "csrc"), "-I", os.path.join(ROOT, "tests", "emu"),
                    os.path.join(ROOT, "tests", "emu", "conv_bx64s2_emu.cpp"), "-o", out], check=True)
    return out


# (16 x 32 -> 8 x 16: one full unit; 30 x 40 -> 15 x 20: partial rows and strips, two units per workgroup; 9 x 11: W % 4 != 0 and odd sizes; cout 128: two cout halves)
@pytest.mark.parametrize("cout,shape,grid", [(64, (1, 16, 32), 1), (64, (2, 30, 40), 3), (128, (1, 30, 40), 2), (64, (1, 9, 11), 1), (128, (1, 18, 22), 2), (64, (3, 60, 80), 4)])
def test_conv_bx64s2_body_on_the_host(emu_bin, cout, shape, grid):
    B, H, W = shape
    g = torch.Generator().manual_seed(100 + cout + H)
    x = torch.randn(B, 64, H, W, generator=g) * 2
    w = torch.randn(cout, 64, 3, 3, generator=g) / 24
    b = torch.randn(cout, generator=g) * 0.3
    blob = np.concatenate([np.array([B, H, W, cout, 1, grid], np.int32).view(np.float32)] + [t.numpy().astype(np.float32).reshape(-1) for t in (x, w, b)])
    out = subprocess.run([emu_bin], input=blob.tobytes(), capture_output=True, check=True, timeout=600).stdout
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1))
    y = np.frombuffer(out[:-4], np.float32).reshape(tuple(ref.shape))
    d = np.abs(y - ref.numpy())
    print(f"cout {cout} {shape}: max |err| {d.max():.3g}, max |y| {float(ref.abs().max()):.3g}")
    assert int(np.frombuffer(out[-4:], np.int32)[0]) == 0 and np.isfinite(y).all()
    assert d.max() <= 3e-6 * float(ref.abs().max())


def test_a_lax_wait_count_is_caught_by_late_dma_delivery():
    """The negative control of the deferred-DMA emulation: the same source with the row barriers' s_waitcnt vmcnt(3) replaced by vmcnt(6) -- TWO weight rows allowed to stay in
    flight, the row being opened among them -- must deliver wrong results when the DMA arrives as late as the waits allow (and right ones when it arrives at issue: the
    at-issue run alone would not have noticed)."""
    if not os.path.exists(CLANG):
        pytest.skip("no host clang")
    B, H, W, cout, grid = 1, 16, 32, 64, 1
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, 64, H, W, generator=g) * 2
    w = torch.randn(cout, 64, 3, 3, generator=g) / 24
    b = torch.randn(cout, generator=g) * 0.3
    blob = np.concatenate([np.array([B, H, W, cout, 1, grid], np.int32).view(np.float32)] + [t.numpy().astype(np.float32).reshape(-1) for t in (x, w, b)])
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1)).numpy()
    errs = {}
    for mode, flags in (("late", ["-DEMU_DEFER_DMA"]), ("at_issue", [])):
        out = os.path.join(tempfile.mkdtemp(), "conv_bx64s2_emu_lax")
        subprocess.run([CLANG, "-O1", "-w", "-std=c++20", "-pthread", "-DXFH_S2_WAIT_PIECES=(2 * NPW)"] + flags + ["-I", os.path.join(ROOT, "accelerated_features_amd", "csrc"), "-I", os.path.join(ROOT, "tests", "emu"),
                        os.path.join(ROOT, "tests", "emu", "conv_bx64s2_emu.cpp"), "-o", out], check=True)
        o = subprocess.run([out], input=blob.tobytes(), capture_output=True, check=True, timeout=600).stdout
        errs[mode] = float(np.abs(np.frombuffer(o[:-4], np.float32).reshape(ref.shape) - ref).max())
    print(errs)
    assert errs["at_issue"] <= 3e-6 * float(np.abs(ref).max()) and errs["late"] > 1e-2


def test_every_kernel_that_claims_an_emulated_dma_protocol_has_one():
    """tools/check_dma_barriers.py skips kernels that carry the marker XFH_DMA_PROTOCOL_EMULATED; the claim must be backed by a test that builds that source with EMU_DEFER_DMA."""
    import glob
    import re
    csrc = os.path.join(ROOT, "accelerated_features_amd", "csrc")
    marked = [os.path.basename(f) for f in sorted(glob.glob(os.path.join(csrc, "*"))) if re.search(r"^\s*XFH_DMA_PROTOCOL_EMULATED\(\);", open(f).read(), re.M)]
    assert marked == ["conv_bx64s2_body.hpp"], marked
    me = open(__file__).read()
    assert "-DEMU_DEFER_DMA" in me and "conv_bx64s2_emu.cpp" in me and '#include "conv_bx64s2_body.hpp"' in open(os.path.join(ROOT, "tests", "emu", "conv_bx64s2_emu.cpp")).read()
