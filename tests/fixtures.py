"""Seeded synthetic fixtures shared by the CPU and GPU tests, ``bench.py`` and ``smoke()``.

The reference ships neither trained weights (``.MISSING_LARGE_BLOBS``) nor test vectors, so
parity is checked on synthetic weights and images (SURVEY.md finding 3/5, App. B):

* convolution / linear weights: numpy ``RandomState`` (bit-reproducible on every machine),
  kaiming-uniform-like bound 1/sqrt(fan_in) times a gain that keeps activations O(1);
* BatchNorm running statistics: CALIBRATED once (``tests/golden/make_golden.py``) so every
  layer's pre-activation is roughly zero-mean/unit-variance like a trained net, and committed
  as ``tests/golden/bn_stats.npz`` (they cannot be regenerated bit-exactly on another
  machine because they depend on the CPU conv summation order);
* ``block_fusion.2`` (the plain 1x1 conv that emits the 64-D descriptors) is composed with a
  ZCA whitening of its own output, calibrated by the same script and committed in the same
  file: whitened descriptors make the raw-dot-product mutual-NN of the semi-dense matcher find
  thousands of matches on a noisy copy of an image (plain random weights: ~1 of 4095, SURVEY
  App. B.3), so ``match_xfeat_star`` has real rows to refine and compare;
* ``keypoint_head.3.weight`` is sharpened so the heat map clears the 0.05 detection
  threshold (default init gives zero keypoints), ``fine_matcher.12.weight`` likewise so the
  refinement confidences clear 0.25.
"""
import os
import sys

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from accelerated_features_amd.spec import CONVS, FINE  # noqa: E402

GOLDEN_DIR = os.path.join(_HERE, "golden")
BN_STATS = os.path.join(GOLDEN_DIR, "bn_stats.npz")

KEYPOINT_SHARPEN = 6.0
FINE_SHARPEN = 1.0


def raw_state_dict(seed=0):
    """Weights from numpy RNG; BN stats = identity (mean 0, var 1)."""
    rs = np.random.RandomState(seed)
    sd = {}
    for c in CONVS:
        fan_in = c.cin * c.k * c.k
        bound = np.sqrt(3.0) * np.sqrt(2.0 / fan_in)      # unit-gain (He) uniform
        w = rs.uniform(-bound, bound, size=(c.cout, c.cin, c.k, c.k)).astype(np.float32)
        if c.kind == "bn":
            sd[f"{c.name}.layer.0.weight"] = torch.from_numpy(w)
            sd[f"{c.name}.layer.1.running_mean"] = torch.zeros(c.cout)
            sd[f"{c.name}.layer.1.running_var"] = torch.ones(c.cout)
            sd[f"{c.name}.layer.1.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
        else:
            b = rs.uniform(-0.1, 0.1, size=(c.cout,)).astype(np.float32)
            if c.name == "keypoint_head.3":
                w = w * np.float32(KEYPOINT_SHARPEN)
            sd[f"{c.name}.weight"] = torch.from_numpy(w)
            sd[f"{c.name}.bias"] = torch.from_numpy(b)
    for li, fin, fout, bi in FINE:
        bound = np.sqrt(3.0) * np.sqrt(2.0 / fin)
        w = rs.uniform(-bound, bound, size=(fout, fin)).astype(np.float32)
        b = rs.uniform(-0.1, 0.1, size=(fout,)).astype(np.float32)
        if li == 12:
            w = w * np.float32(FINE_SHARPEN)
        sd[f"fine_matcher.{li}.weight"] = torch.from_numpy(w)
        sd[f"fine_matcher.{li}.bias"] = torch.from_numpy(b)
        if bi is not None:
            sd[f"fine_matcher.{bi}.running_mean"] = torch.zeros(fout)
            sd[f"fine_matcher.{bi}.running_var"] = torch.ones(fout)
            sd[f"fine_matcher.{bi}.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    return sd


def synthetic_state_dict(seed=0):
    """The fixture every test uses: raw weights + committed calibrated BN statistics."""
    sd = raw_state_dict(seed)
    if not os.path.exists(BN_STATS):
        raise FileNotFoundError(f"{BN_STATS} missing: run tests/golden/make_golden.py")
    st = np.load(BN_STATS)
    for k in st.files:
        assert k in sd, k
        sd[k] = torch.from_numpy(st[k].astype(np.float32))
    return sd


def _smooth_noise(rs, H, W, s):
    h, w = -(-H // s), -(-W // s)
    n = rs.rand(h + 2, w + 2).astype(np.float32)
    up = np.kron(n, np.ones((s, s), np.float32))
    if s > 1:     # separable box blur of width s -> piecewise-linear interpolation of the grid
        k = np.ones(s, np.float32) / s
        up = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 0, up)
        up = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, up)
    return up[s : s + H, s : s + W]


def texture_images(B, H, W, seed=7, channels=3):
    """Band-limited texture in [0,1]: sum of smooth noise at scales 1,2,4,8,16."""
    rs = np.random.RandomState(seed)
    out = np.empty((B, channels, H, W), np.float32)
    for b in range(B):
        for c in range(channels):
            acc = np.zeros((H, W), np.float32)
            for s in (1, 2, 4, 8, 16):
                acc += _smooth_noise(rs, H, W, s) * np.float32(np.sqrt(s))
            lo, hi = acc.min(), acc.max()
            out[b, c] = (acc - lo) / (hi - lo)
    return torch.from_numpy(out)


def shifted_pair(B, H, W, seed=7, shift=(16, 24), noise=0.02):
    """(a, b): b = roll(a, shift) + small noise -- a pair with genuine correspondences."""
    a = texture_images(B, H, W, seed)
    rs = np.random.RandomState(seed + 1000)
    b = torch.roll(a, shifts=shift, dims=(2, 3)) + torch.from_numpy(
        (noise * rs.randn(B, 3, H, W)).astype(np.float32))
    return a, b


def star_pair(B, H, W, seed=41, noise=0.005):
    """(a, b): b = a + small noise.  The pair the semi-dense tests use: the dual-scale path resizes by 0.6 and 1.3 and
    the backbone strides by 32, so no non-trivial shift keeps both scales cell-aligned; a noisy copy gives thousands
    of mutual matches whose refinement (fine_matcher offsets, confidence filter) is real work to compare."""
    return shifted_pair(B, H, W, seed=seed, shift=(0, 0), noise=noise)


# ------------------------------------------------------------------------------------------------------------
# LighterGlue: synthetic weights in the key layout kornia's LightGlue has after the reference's loader
# (modules/lighterglue.py:41-48); the trained xfeat-lighterglue.pt is not in the snapshot.
# ------------------------------------------------------------------------------------------------------------
def lighterglue_state_dict(seed=0):
    import numpy as np
    import torch
    from oracle.lighterglue_oracle import state_dict_keys
    rs = np.random.RandomState(1000 + seed)
    sd = {}
    for name, shape in state_dict_keys():
        if name.endswith("ffn.1.weight"):                       # LayerNorm gain
            v = 1.0 + 0.1 * rs.randn(*shape)
        elif name.endswith("ffn.1.bias"):
            v = 0.1 * rs.randn(*shape)
        elif name == "posenc.Wr.weight":                        # frequencies: a few radians over the image
            v = 2.0 * rs.randn(*shape)
        elif name.endswith("matchability.bias"):
            v = 0.5 + 0.1 * rs.randn(*shape)                    # most points stay matchable: pruning removes a minority
        elif name.endswith(".bias"):
            v = 0.05 * rs.randn(*shape)
        elif name.endswith("matchability.weight") or name.endswith("token.0.weight"):
            v = rs.randn(*shape) * (1.5 / np.sqrt(shape[-1]))
        else:                                                   # Linear weights (out, in)
            v = rs.randn(*shape) * (1.0 / np.sqrt(shape[-1]))
        sd[name] = torch.from_numpy(v.astype(np.float32))
    return sd


def lighterglue_inputs(n0, n1, seed=0, size0=(640, 480), size1=(640, 480)):
    """Two sets of key-points (pixels) with unit-norm 64-D descriptors; a third of set 1 are noisy copies of set 0."""
    import numpy as np
    import torch
    rs = np.random.RandomState(2000 + seed)
    k0 = np.stack([rs.uniform(0, size0[0] - 1, n0), rs.uniform(0, size0[1] - 1, n0)], -1).astype(np.float32)
    k1 = np.stack([rs.uniform(0, size1[0] - 1, n1), rs.uniform(0, size1[1] - 1, n1)], -1).astype(np.float32)
    d0 = rs.randn(n0, 64).astype(np.float32)
    d1 = rs.randn(n1, 64).astype(np.float32)
    m = min(n0, n1) // 3
    perm = rs.permutation(n1)[:m]
    d1[perm] = d0[:m] + 0.05 * rs.randn(m, 64).astype(np.float32)
    k1[perm] = np.clip(k0[:m] + rs.randn(m, 2).astype(np.float32) * 3, 0, [size1[0] - 1, size1[1] - 1])
    d0 /= np.linalg.norm(d0, axis=1, keepdims=True)
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    return (torch.from_numpy(k0), torch.from_numpy(d0), torch.tensor(size0, dtype=torch.float32),
            torch.from_numpy(k1), torch.from_numpy(d1), torch.tensor(size1, dtype=torch.float32))
