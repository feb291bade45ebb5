"""The explicit fp32 sampling arithmetic of the oracle (the contract the HIP kernels follow,
SURVEY App. A.6) against torch's grid_sample, exactly as InterpolateSparse2d calls it
(/root/reference/modules/interpolator.py:17-32)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import xfeat_oracle as O


def _ref_sample(m, pos, H, W, mode):
    grid = 2.0 * (pos / torch.tensor([W - 1, H - 1], dtype=torch.float32)) - 1.0
    out = F.grid_sample(m[None], grid[None, :, None, :].float(), mode=mode, align_corners=False)
    return out[0, :, :, 0].t()


@pytest.mark.parametrize("H,W", [(480, 640), (96, 128), (576, 800), (608, 608), (1312, 1312)])
def test_nearest_maps_every_pixel_to_itself_except_last_row_col(H, W):
    m = torch.arange(H * W, dtype=torch.float32).reshape(1, H, W) + 1
    xs = torch.arange(W)
    pos = torch.stack([xs, torch.full_like(xs, 3)], -1)
    v = O.sample_nearest(m, pos, H, W)[:, 0]
    exp = m[0, 3, :].clone()
    exp[-1] = 0
    assert torch.equal(v, exp)
    assert torch.equal(v, _ref_sample(m, pos, H, W, "nearest")[:, 0])
    ys = torch.arange(H)
    pos = torch.stack([torch.full_like(ys, 5), ys], -1)
    v = O.sample_nearest(m, pos, H, W)[:, 0]
    exp = m[0, :, 5].clone()
    exp[-1] = 0
    assert torch.equal(v, exp)
    assert torch.equal(v, _ref_sample(m, pos, H, W, "nearest")[:, 0])


@pytest.mark.parametrize("mode,fn,tol", [("bilinear", O.sample_bilinear, 2e-6), ("bicubic", O.sample_bicubic, 5e-6)])
def test_bilinear_bicubic_match_grid_sample(mode, fn, tol):
    g = torch.Generator().manual_seed(0)
    H, W, h, w = 480, 640, 60, 80
    m = torch.randn(7, h, w, generator=g)
    pos = torch.stack([torch.randint(0, W, (4000,), generator=g), torch.randint(0, H, (4000,), generator=g)], -1)
    corners = torch.tensor([[0, 0], [W - 1, 0], [0, H - 1], [W - 1, H - 1], [W - 2, H - 2], [1, 1], [7, 7], [8, 8]])
    pos = torch.cat([corners, pos])
    a = fn(m, pos, H, W)
    b = _ref_sample(m, pos, H, W, mode)
    assert float((a - b).abs().max()) <= tol


def test_heatmap_layout_and_unfold_order():
    g = torch.Generator().manual_seed(1)
    logits = torch.randn(2, 65, 3, 4, generator=g)
    heat = O.kpts_heatmap(logits)
    p = torch.softmax(logits, 1)
    for (b, i, j, dy, dx) in [(0, 0, 0, 0, 0), (1, 2, 3, 7, 6), (0, 1, 2, 3, 5)]:
        assert heat[b, 0, 8 * i + dy, 8 * j + dx] == p[b, 8 * dy + dx, i, j]
    x = torch.randn(1, 1, 16, 24, generator=g)
    u = O.unfold8(x)
    for (i, j, dy, dx) in [(0, 0, 0, 0), (1, 2, 7, 6), (1, 0, 3, 5)]:
        assert u[0, 8 * dy + dx, i, j] == x[0, 0, 8 * i + dy, 8 * j + dx]


def test_nms_plateau_threshold_and_order():
    heat = torch.zeros(1, 1, 16, 16)
    heat[0, 0, 3, 3] = 0.5
    heat[0, 0, 3, 4] = 0.5          # plateau: both kept
    heat[0, 0, 10, 2] = 0.05        # == threshold: strict >, dropped
    heat[0, 0, 12, 12] = 0.9
    heat[0, 0, 12, 14] = 0.8        # inside the 5x5 window of a larger value: dropped
    heat[0, 0, 0, 15] = 0.3
    k = O.nms(heat, 0.05, 5)[0].tolist()
    assert k == [[15, 0], [3, 3], [4, 3], [12, 12]]


def test_subpix_one_hot():
    o = torch.full((1, 64), -50.0)
    o[0, 8 * 2 + 5] = 50.0
    assert torch.allclose(O.subpix_softmax2d(o), torch.tensor([[1.0, -2.0]]))
