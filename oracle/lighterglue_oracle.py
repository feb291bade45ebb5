"""CPU oracle of LighterGlue -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; nothing under
accelerated_features_amd/ does.

What it restates.  The reference does not contain the arithmetic: `modules/lighterglue.py:7-57` configures
`kornia.feature.lightglue.LightGlue` (kornia==0.7.2, `requirements.txt:3`; not vendored, not installed) with
`default_conf_xfeat` (`modules/lighterglue.py:12-27`: input_dim 64, descriptor_dim 96, 6 layers, 1 head,
depth_confidence -1, width_confidence 0.95, filter_threshold = min_conf) and `modules/xfeat.py:131-162` calls it
on one pair.  kornia 0.7.2's module is the published LightGlue v0.1 architecture (Lindenberger et al., ICCV'23);
this file restates that published algorithm, with the state_dict key names the reference's loader ends up with
(`modules/lighterglue.py:41-48`).

PARITY STATUS: **unpinned against kornia 0.7.2** (absent).  What IS pinned: every building block (rotary
self-attention block, bidirectional cross-attention block, FFN, match assignment, mutual filter, key-point
normalisation, positional encoding) against the independent HuggingFace port
(`transformers/models/lightglue/modeling_lightglue.py`) with tied weights, AND the whole matcher end to end without
width pruning against that port's own `LightGlueForKeypointMatching._match_image_pair` (identical match list, scores
within 2e-5) -- see tests/golden/make_golden_lighterglue.py and tests/test_oracle_lighterglue.py.  Not pinned by anything on disk:
the width-pruning rule and its CPU threshold (`pruning_keypoint_thresholds['cpu'] = -1`, i.e. pruning is always
on for CPU tensors), restated from the published implementation.
"""
import math

import torch
import torch.nn.functional as F

CONF = dict(input_dim=64, descriptor_dim=96, n_layers=6, num_heads=1, depth_confidence=-1.0, width_confidence=0.95,
            filter_threshold=0.1)


def state_dict_keys(conf=CONF):
    """(name, shape) of every tensor of kornia's LightGlue under the reference's configuration, in module order."""
    d, n = conf["descriptor_dim"], conf["n_layers"]
    keys = [("input_proj.weight", (d, conf["input_dim"])), ("input_proj.bias", (d,)), ("posenc.Wr.weight", (d // conf["num_heads"] // 2, 2))]
    for i in range(n):
        p = f"transformers.{i}.self_attn."
        keys += [(p + "Wqkv.weight", (3 * d, d)), (p + "Wqkv.bias", (3 * d,)), (p + "out_proj.weight", (d, d)), (p + "out_proj.bias", (d,)),
                 (p + "ffn.0.weight", (2 * d, 2 * d)), (p + "ffn.0.bias", (2 * d,)), (p + "ffn.1.weight", (2 * d,)), (p + "ffn.1.bias", (2 * d,)),
                 (p + "ffn.3.weight", (d, 2 * d)), (p + "ffn.3.bias", (d,))]
        p = f"transformers.{i}.cross_attn."
        keys += [(p + "to_qk.weight", (d, d)), (p + "to_qk.bias", (d,)), (p + "to_v.weight", (d, d)), (p + "to_v.bias", (d,)),
                 (p + "to_out.weight", (d, d)), (p + "to_out.bias", (d,)),
                 (p + "ffn.0.weight", (2 * d, 2 * d)), (p + "ffn.0.bias", (2 * d,)), (p + "ffn.1.weight", (2 * d,)), (p + "ffn.1.bias", (2 * d,)),
                 (p + "ffn.3.weight", (d, 2 * d)), (p + "ffn.3.bias", (d,))]
    for i in range(n):
        p = f"log_assignment.{i}."
        keys += [(p + "matchability.weight", (1, d)), (p + "matchability.bias", (1,)), (p + "final_proj.weight", (d, d)), (p + "final_proj.bias", (d,))]
    for i in range(n - 1):
        keys += [(f"token_confidence.{i}.token.0.weight", (1, d)), (f"token_confidence.{i}.token.0.bias", (1,))]
    return keys


# ------------------------------------------------------------------------------------------------------------
# building blocks (single pair, no batch dimension: the reference only supports B = 1, modules/xfeat.py:134)
# ------------------------------------------------------------------------------------------------------------
def normalize_keypoints(kpts, size):
    """kpts (N,2) pixels, size (2,) = (W,H): (kp - size/2) / (max(W,H)/2)."""
    size = size.to(kpts.dtype)
    return (kpts - size / 2) / (size.max() / 2)


def posenc(sd, kpts_n):
    """Learnable Fourier encoding -> (cos, sin), each (N, d) with every frequency repeated twice (interleaved)."""
    proj = kpts_n @ sd["posenc.Wr.weight"].t()
    return torch.cos(proj).repeat_interleave(2, dim=-1), torch.sin(proj).repeat_interleave(2, dim=-1)


def rotate_half(x):
    x1, x2 = x[..., 0::2], x[..., 1::2]
    return torch.stack((-x2, x1), dim=-1).flatten(-2)


def apply_rotary(freqs, t):
    return t * freqs[0] + rotate_half(t) * freqs[1]


def ffn(sd, p, x):
    h = F.linear(x, sd[p + "ffn.0.weight"], sd[p + "ffn.0.bias"])
    h = F.layer_norm(h, (h.shape[-1],), sd[p + "ffn.1.weight"], sd[p + "ffn.1.bias"], 1e-5)
    h = F.gelu(h)
    return F.linear(h, sd[p + "ffn.3.weight"], sd[p + "ffn.3.bias"])


def self_block(sd, i, x, enc):
    """x (N,d).  One fused Wqkv; its output feature 3*c + t is component c of (q,k,v)[t] (unflatten(-1,(h,-1,3)))."""
    p = f"transformers.{i}.self_attn."
    d = x.shape[-1]
    qkv = F.linear(x, sd[p + "Wqkv.weight"], sd[p + "Wqkv.bias"]).reshape(-1, d, 3)
    q, k, v = apply_rotary(enc, qkv[..., 0]), apply_rotary(enc, qkv[..., 1]), qkv[..., 2]
    attn = torch.softmax((q @ k.t()) * (d ** -0.5), dim=-1)
    msg = F.linear(attn @ v, sd[p + "out_proj.weight"], sd[p + "out_proj.bias"])
    return x + ffn(sd, p, torch.cat([x, msg], -1))


def cross_block(sd, i, x0, x1):
    """Shared to_qk projection, one similarity matrix, softmax along rows for image 0 and along columns for image 1."""
    p = f"transformers.{i}.cross_attn."
    d = x0.shape[-1]
    qk0, qk1 = F.linear(x0, sd[p + "to_qk.weight"], sd[p + "to_qk.bias"]), F.linear(x1, sd[p + "to_qk.weight"], sd[p + "to_qk.bias"])
    v0, v1 = F.linear(x0, sd[p + "to_v.weight"], sd[p + "to_v.bias"]), F.linear(x1, sd[p + "to_v.weight"], sd[p + "to_v.bias"])
    s = d ** -0.25
    sim = (qk0 * s) @ (qk1 * s).t()
    m0 = torch.softmax(sim, dim=-1) @ v1
    m1 = torch.softmax(sim.t(), dim=-1) @ v0
    m0 = F.linear(m0, sd[p + "to_out.weight"], sd[p + "to_out.bias"])
    m1 = F.linear(m1, sd[p + "to_out.weight"], sd[p + "to_out.bias"])
    return x0 + ffn(sd, p, torch.cat([x0, m0], -1)), x1 + ffn(sd, p, torch.cat([x1, m1], -1))


def transformer_layer(sd, i, d0, d1, e0, e1):
    d0, d1 = self_block(sd, i, d0, e0), self_block(sd, i, d1, e1)
    return cross_block(sd, i, d0, d1)


def matchability(sd, i, x):
    return torch.sigmoid(F.linear(x, sd[f"log_assignment.{i}.matchability.weight"], sd[f"log_assignment.{i}.matchability.bias"])).squeeze(-1)


def log_assignment(sd, i, d0, d1):
    """(M+1, N+1) log assignment matrix with dustbin row / column."""
    p = f"log_assignment.{i}."
    d = d0.shape[-1]
    md0 = F.linear(d0, sd[p + "final_proj.weight"], sd[p + "final_proj.bias"]) / d ** 0.25
    md1 = F.linear(d1, sd[p + "final_proj.weight"], sd[p + "final_proj.bias"]) / d ** 0.25
    sim = md0 @ md1.t()
    z0 = F.linear(d0, sd[p + "matchability.weight"], sd[p + "matchability.bias"])      # (M,1)
    z1 = F.linear(d1, sd[p + "matchability.weight"], sd[p + "matchability.bias"])      # (N,1)
    m, n = sim.shape
    cert = F.logsigmoid(z0) + F.logsigmoid(z1).t()
    scores = sim.new_zeros((m + 1, n + 1))
    scores[:m, :n] = F.log_softmax(sim, 1) + F.log_softmax(sim.t().contiguous(), 1).t() + cert
    scores[:m, n] = F.logsigmoid(-z0.squeeze(-1))
    scores[m, :n] = F.logsigmoid(-z1.squeeze(-1))
    return scores


def filter_matches(scores, th):
    """mutual arg-max on the (M,N) core + exp(score) > th.  Returns m0 (M,) with -1 for no match, mscores0 (M,)."""
    core = scores[:-1, :-1]
    max0, max1 = core.max(1), core.max(0)
    m0, m1 = max0.indices, max1.indices
    mutual0 = torch.arange(m0.shape[0]) == m1[m0]
    ms0 = torch.where(mutual0, max0.values.exp(), torch.zeros(()))
    valid0 = mutual0 & (ms0 > th)
    return torch.where(valid0, m0, torch.full_like(m0, -1)), ms0


# ------------------------------------------------------------------------------------------------------------
# the matcher (kornia LightGlue.forward under the reference's conf; one pair)
# ------------------------------------------------------------------------------------------------------------
def lighterglue_forward(sd, kpts0, desc0, size0, kpts1, desc1, size1, min_conf=0.1, conf=CONF, prune=True, trace=None, prune_min_kpts=-1):
    """kpts (N,2) pixel coordinates, desc (N,64), size (2,) = (W,H).
    Returns matches (S,2) int64 (indices into the ORIGINAL key-point lists, ascending in column 0) and scores (S,).
    `prune=True`: width pruning (matchability > 1 - width_confidence) after every layer but the last, applied to a set
    while it holds more than `prune_min_kpts` points (published pruning_keypoint_thresholds: cpu -1 = always,
    cuda 1024, flash 1536)."""
    n_layers = conf["n_layers"]
    k0, k1 = normalize_keypoints(kpts0, size0), normalize_keypoints(kpts1, size1)
    d0 = F.linear(desc0, sd["input_proj.weight"], sd["input_proj.bias"])
    d1 = F.linear(desc1, sd["input_proj.weight"], sd["input_proj.bias"])
    e0, e1 = posenc(sd, k0), posenc(sd, k1)
    do_prune = prune and conf["width_confidence"] > 0
    ind0, ind1 = torch.arange(d0.shape[0]), torch.arange(d1.shape[0])
    for i in range(n_layers):
        d0, d1 = transformer_layer(sd, i, d0, d1, e0, e1)
        if trace is not None:
            trace.append((d0.clone(), d1.clone(), ind0.clone(), ind1.clone()))
        if i == n_layers - 1:
            continue
        # depth_confidence = -1: no early stop, hence no token confidences in the pruning rule
        if do_prune and d0.shape[0] > prune_min_kpts:
            keep0 = torch.where(matchability(sd, i, d0) > 1 - conf["width_confidence"])[0]
            ind0, d0, e0 = ind0[keep0], d0[keep0], (e0[0][keep0], e0[1][keep0])
        if do_prune and d1.shape[0] > prune_min_kpts:
            keep1 = torch.where(matchability(sd, i, d1) > 1 - conf["width_confidence"])[0]
            ind1, d1, e1 = ind1[keep1], d1[keep1], (e1[0][keep1], e1[1][keep1])
    if d0.shape[0] == 0 or d1.shape[0] == 0:
        return torch.zeros((0, 2), dtype=torch.int64), torch.zeros((0,))
    scores = log_assignment(sd, n_layers - 1, d0, d1)
    if isinstance(trace, list):
        trace.append((scores, None, ind0.clone(), ind1.clone()))          # last entry: the (M+1, N+1) log assignment
    m0, ms0 = filter_matches(scores, min_conf)
    valid = m0 > -1
    a = torch.where(valid)[0]
    b = m0[valid]
    return torch.stack([ind0[a], ind1[b]], -1), ms0[valid]
