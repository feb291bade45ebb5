"""The fine_matcher's layers in the fp16-pair arithmetic (csrc/linear_fx_body.hpp: linear_fx_body -- fp32 / gathered / split-form input, fp32 / split-form output, LDS staging
through registers -- and linear_fxd_body -- the chain's inner layers, every operand by LDS-DMA into a swizzled stage) compiled for the HOST (tests/emu/) against a float64
product.  Checks the index arithmetic (staging, swizzle, fragment order, the transposed epilogue), the barrier structure and the live-row handling; the split form y = yh + 2^-11 yl
is checked value by value (modules/model.py:97-111 is the layer stack; the GPU parity tests of match_xfeat_star run the kernels themselves)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
TOL = 2e-6      # of max |y|: the pair arithmetic is fp32-equivalent (bx_split.hpp: 2.7e-7 measured on K = 576 sums)


@pytest.fixture(scope="module")
def emu_bin():
    if not os.path.exists(CLANG):
        pytest.skip("no host clang")
    out = os.path.join(tempfile.mkdtemp(), "linear_fx_emu")
    subprocess.run([CLANG, "-O1", "-w", "-std=c++20", "-pthread", "-I", os.path.join(ROOT, "accelerated_features_amd", "csrc"), "-I", os.path.join(ROOT, "tests", "emu"),
                    os.path.join(ROOT, "tests", "emu", "linear_fx_emu.cpp"), "-o", out], check=True)
    return out


def _blob(hdr, arrs):
    return np.concatenate([np.array(hdr, np.int32).view(np.float32)] + [np.asarray(a, np.float32).reshape(-1) for a in arrs]).tobytes()


def _run(emu_bin, M, K, N, inn, out, relu, mlive, seed=0, scale=3.0, poke=None):
    rng = np.random.default_rng(seed)
    n_pad = N
    w = (rng.standard_normal((K, N)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    if inn == 2:      # the gather of two descriptors: the harness pairs row r of desc0 with row M - 1 - r of desc1
        d0 = rng.standard_normal((M, 64)).astype(np.float32)
        d1 = rng.standard_normal((M, 64)).astype(np.float32)
        x = np.concatenate([d0, d1[::-1]], 1)
        arrs = [d0, d1, w, b]
    else:
        x = (rng.standard_normal((M, K)) * scale).astype(np.float32)
        if poke is not None:
            x[poke[0], poke[1]] = poke[2]
        arrs = [x, w, b]
    o = subprocess.run([emu_bin], input=_blob([M, K, N, n_pad, inn, out, relu, mlive], arrs), capture_output=True, check=True, timeout=900).stdout
    live = M if mlive < 0 else min(M, mlive)
    ref = x.astype(np.float64) @ w.astype(np.float64) + b
    if relu:
        ref = np.maximum(ref, 0)
    if out == 0:
        y = np.frombuffer(o[:M * N * 4], np.float32).reshape(M, N)
        status = int(np.frombuffer(o[M * N * 4:], np.int32)[0])
        return y[:live], ref[:live], status, bool(np.isnan(y[live:]).all())
    yp = np.frombuffer(o[:M * 2 * n_pad * 2], np.uint16).reshape(M, 2, n_pad)
    status = int(np.frombuffer(o[M * 2 * n_pad * 2:], np.int32)[0])
    val = yp[:, 0].view(np.float16).astype(np.float64) + yp[:, 1].view(np.float16).astype(np.float64) / 2048
    return val[:live], ref[:live], status, bool((yp[live:] == 0x7e00).all())


# (in: 0 fp32 rows, 2 the gather, 3 the split form, 4 the split form through the DMA kernel; out: 0 fp32, 1 the split form; mlive: the device-side live row count, -1 = none)
@pytest.mark.parametrize("M,K,N,inn,out,relu,mlive", [
    (300, 128, 128, 0, 1, 1, -1),       # the chain's first layer from fp32 rows (xfh_fine_matcher)
    (300, 128, 64, 2, 1, 1, 290),       # ... from the gather of the two descriptors (xfh_refine_matches), a live count inside the second row block
    (300, 512, 64, 3, 0, 0, 257),       # the last layer: split form in, fp32 out
    (520, 512, 128, 3, 1, 1, -1),       # an inner layer through the register-staged form
    (520, 512, 128, 4, 1, 1, -1),       # ... through the DMA form: three row blocks, the last one 8 rows
    (300, 512, 256, 4, 1, 1, 270),      # two column blocks of 128, a live count
    (256, 128, 128, 4, 1, 0, -1),       # the shortest K the three-stage ring takes (four chunks)
    (100, 64, 64, 0, 0, 0, -1),         # two chunks: the shortest K of the register-staged form
    (1, 128, 64, 2, 1, 1, -1),          # one match (every staged row but one is the clamped last live row)
    (300, 512, 128, 4, 1, 1, 1),        # one live row of 300: the second row block leaves, the DMA writes zeros behind row 0
    (300, 512, 128, 4, 1, 1, 0),        # no live row: every workgroup leaves, nothing is written
    (300, 512, 64, 3, 0, 1, 0),
])
def test_linear_fx_bodies_on_the_host(emu_bin, M, K, N, inn, out, relu, mlive):
    y, ref, status, untouched = _run(emu_bin, M, K, N, inn, out, relu, mlive)
    if mlive == 0:
        assert status == 0 and untouched and y.shape[0] == 0
        return
    err = np.abs(y - ref).max() / np.abs(ref).max()
    print(f"M {M} K {K} N {N} in {inn} out {out} live {mlive}: max |err| / max |y| = {err:.2e}")
    assert status == 0 and untouched
    assert err < TOL


def test_linear_fx_reports_its_range(emu_bin):
    """an fp32 input beyond the fp16 range sets bit 0 of the status word (the host re-runs the chain on the f32 matrix cores)"""
    *_, status, _ = _run(emu_bin, 300, 128, 128, 0, 1, 1, -1, poke=(17, 5, 7e4))
    assert status == 1
