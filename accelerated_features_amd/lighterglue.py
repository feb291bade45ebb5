"""MI355X-native LighterGlue, drop-in for the reference's ``modules/lighterglue.py::LighterGlue``.

The reference class wraps ``kornia.feature.lightglue.LightGlue`` configured with ``default_conf_xfeat``
(/root/reference/modules/lighterglue.py:12-27) and is called by ``XFeat.match_lighterglue``
(/root/reference/modules/xfeat.py:131-162).  Here the whole matcher runs in HIP kernels behind
``xfh_lg_match`` (include/xfeat_hip.h); this module only holds the weights, the key renaming of the reference's
loader, and the workspace.  No kornia, no second backend: without a GPU or the shared library every call raises.
"""
import ctypes as C
import os

import numpy as np
import torch
from torch import nn

from . import _lib

__all__ = ["LighterGlue"]


def _keys(conf):
    """(name, shape) in the order xfh_lg_create expects = kornia's module order."""
    d, n = conf["descriptor_dim"], conf["n_layers"]
    ffn = lambda p: [(p + "ffn.0.weight", (2 * d, 2 * d)), (p + "ffn.0.bias", (2 * d,)), (p + "ffn.1.weight", (2 * d,)),
                     (p + "ffn.1.bias", (2 * d,)), (p + "ffn.3.weight", (d, 2 * d)), (p + "ffn.3.bias", (d,))]
    lin = lambda p, o=d: [(p + ".weight", (o, d)), (p + ".bias", (o,))]
    keys = [("input_proj.weight", (d, conf["input_dim"])), ("input_proj.bias", (d,)), ("posenc.Wr.weight", (d // 2, 2))]
    for i in range(n):
        p = f"transformers.{i}.self_attn."
        keys += lin(p + "Wqkv", 3 * d) + lin(p + "out_proj") + ffn(p)
        p = f"transformers.{i}.cross_attn."
        keys += lin(p + "to_qk") + lin(p + "to_v") + lin(p + "to_out") + ffn(p)
    for i in range(n):
        keys += lin(f"log_assignment.{i}.matchability", 1) + lin(f"log_assignment.{i}.final_proj")
    for i in range(n - 1):
        keys += lin(f"token_confidence.{i}.token.0", 1)
    return keys


class LighterGlue(nn.Module):
    """
        Lighter version of LightGlue :)  -- same constructor / forward contract as the reference class.
    """

    default_conf_xfeat = {
        "name": "xfeat", "input_dim": 64, "descriptor_dim": 96, "add_scale_ori": False, "add_laf": False, "scale_coef": 1.0,
        "n_layers": 6, "num_heads": 1, "flash": True, "mp": False, "depth_confidence": -1, "width_confidence": 0.95,
        "filter_threshold": 0.1, "weights": None,
    }
    # the published code prunes a set only while it holds more points than this (per device type); the reference runs
    # with flash=True on a GPU, so 'flash' is what its users get and what this class uses
    pruning_keypoint_thresholds = {"cpu": -1, "cuda": 1024, "flash": 1536}

    def __init__(self, weights=os.path.abspath(os.path.dirname(__file__)) + '/../weights/xfeat-lighterglue.pt'):
        super().__init__()
        self.conf = dict(self.default_conf_xfeat)
        self.dev = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        self.prune_min_kpts = self.pruning_keypoint_thresholds["flash"] if self.conf["width_confidence"] > 0 else _lib.LG_NO_PRUNING
        for name, shape in _keys(self.conf):
            self.register_buffer(name.replace(".", "__"), torch.zeros(shape, dtype=torch.float32))
        self._handle, self._handle_device, self._ws = None, None, None
        if weights is not None:
            if isinstance(weights, str):
                if not os.path.exists(weights):
                    raise FileNotFoundError(f"{weights} not found (the reference would download it; this build has no network path)")
                weights = torch.load(weights, map_location='cpu')
            self.load_state_dict(weights)

    # -- weights -------------------------------------------------------------------------------
    @staticmethod
    def rename(state_dict, n_layers=6):
        """The reference loader's renaming of old checkpoint entries (modules/lighterglue.py:41-46)."""
        for i in range(n_layers):
            state_dict = {k.replace(f"self_attn.{i}", f"transformers.{i}.self_attn"): v for k, v in state_dict.items()}
            state_dict = {k.replace(f"cross_attn.{i}", f"transformers.{i}.cross_attn"): v for k, v in state_dict.items()}
            state_dict = {k.replace('matcher.', ''): v for k, v in state_dict.items()}
        return state_dict

    def state_dict(self, *a, **kw):
        return {name: getattr(self, name.replace(".", "__")) for name, _ in _keys(self.conf)}

    def load_state_dict(self, state_dict, strict=False, **kw):
        """strict=False like the reference (extra keys ignored, missing keys keep their initial value)."""
        sd = self.rename(dict(state_dict), self.conf["n_layers"])
        missing = []
        for name, shape in _keys(self.conf):
            if name not in sd:
                missing.append(name)
                continue
            t = torch.as_tensor(sd[name]).detach().to("cpu", torch.float32)
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {name}: {tuple(t.shape)} vs {tuple(shape)}")
            getattr(self, name.replace(".", "__")).copy_(t)
        if strict and missing:
            raise RuntimeError(f"missing keys: {missing}")
        self._drop_handle()
        return missing

    def _drop_handle(self):
        if getattr(self, "_handle", None):
            _lib.load().xfh_lg_destroy(self._handle)
        self._handle = None

    def __del__(self):
        try:
            self._drop_handle()
        except Exception:
            pass

    def handle(self):
        if not torch.cuda.is_available():
            raise _lib.XFeatHipError("accelerated_features_amd needs an AMD MI355X (gfx950) GPU: "
                                     "torch.cuda.is_available() is False and there is no CPU fallback")
        dev = torch.cuda.current_device()
        if self._handle is not None and self._handle_device == dev:
            return self._handle
        self._drop_handle()
        lib = _lib.load()
        sd = self.state_dict()
        arrs = [np.ascontiguousarray(sd[k].detach().to("cpu", torch.float32).numpy()) for k, _ in _keys(self.conf)]
        n = lib.xfh_lg_num_weight_arrays()
        if len(arrs) != n:
            raise _lib.XFeatHipError(f"LighterGlue weight table has {len(arrs)} arrays, library expects {n}")
        for i, a in enumerate(arrs):
            if a.size != lib.xfh_lg_weight_array_floats(i):
                raise _lib.XFeatHipError(f"LighterGlue weight array {i} has {a.size} floats, expected {lib.xfh_lg_weight_array_floats(i)}")
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        h = C.c_void_p()
        _lib.check(lib.xfh_lg_create(ptrs, n, dev, C.byref(h)), "xfh_lg_create")
        self._handle, self._handle_device = h, dev
        return h

    # -- one pair ------------------------------------------------------------------------------
    def match_device(self, kpts0, desc0, size0, kpts1, desc1, size1, min_conf=0.1, prune_min_kpts=None):
        """kpts (N,2), desc (N,64) CUDA tensors, size = (W,H).  Returns device tensors: matches (cap,2) int64, scores (cap,)
        and the int32 count -- no host synchronisation."""
        h = self.handle()
        lib = _lib.load()
        dev = torch.device("cuda", torch.cuda.current_device())
        f = lambda t: t.to(dev, torch.float32).contiguous()
        kpts0, desc0, kpts1, desc1 = f(kpts0), f(desc0), f(kpts1), f(desc1)
        n0, n1 = kpts0.shape[0], kpts1.shape[0]
        if desc0.shape != (n0, 64) or desc1.shape != (n1, 64) or kpts0.shape != (n0, 2) or kpts1.shape != (n1, 2):
            raise ValueError("expected keypoints (N,2) and descriptors (N,64)")
        cap = max(min(n0, n1), 1)
        matches = torch.empty((cap, 2), dtype=torch.int64, device=dev)
        scores = torch.empty((cap,), dtype=torch.float32, device=dev)
        count = torch.zeros((1,), dtype=torch.int32, device=dev)
        if n0 == 0 or n1 == 0:
            return matches, scores, count
        need = lib.xfh_lg_workspace_bytes(n0, n1)
        if self._ws is None or self._ws.numel() < need + 256 or self._ws.device != dev:
            self._ws = torch.empty(int(need) + 256, dtype=torch.uint8, device=dev)
        ws = self._ws[(-self._ws.data_ptr()) % 256:]
        p = lambda t: C.c_void_p(t.data_ptr())
        pm = self.prune_min_kpts if prune_min_kpts is None else prune_min_kpts
        _lib.check(lib.xfh_lg_match(h, p(kpts0), p(desc0), n0, float(size0[0]), float(size0[1]), p(kpts1), p(desc1), n1, float(size1[0]),
                                    float(size1[1]), float(min_conf), int(pm), p(matches), p(scores), p(count), p(ws), need,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)), "xfh_lg_match")
        return matches, scores, count

    def match_pairs_device(self, kpts, desc, counts, size, min_conf=0.1, prune_min_kpts=None):
        """Fixed-capacity batch (the layout XFeat._detect_device produces): kpts (2P,cap,2), desc (2P,cap,64), counts (2P,)
        int32 on the GPU; frames (2p, 2p+1) form pair p; size = (W,H) of every image.  Returns device tensors
        matches (P,cap,2) int64, scores (P,cap), n (P,) int32 without reading the counts back."""
        h = self.handle()
        lib = _lib.load()
        dev = kpts.device
        B, cap = kpts.shape[0], kpts.shape[1]
        if B % 2 or desc.shape != (B, cap, 64) or counts.shape != (B,) or counts.dtype != torch.int32:
            raise ValueError("expected kpts (2P,cap,2), desc (2P,cap,64), counts (2P,) int32")
        kpts, desc = kpts.to(torch.float32).contiguous(), desc.to(torch.float32).contiguous()
        P = B // 2
        matches = torch.empty((P, cap, 2), dtype=torch.int64, device=dev)
        scores = torch.empty((P, cap), dtype=torch.float32, device=dev)
        n = torch.zeros((P,), dtype=torch.int32, device=dev)
        need = lib.xfh_lg_workspace_bytes(cap, cap)
        if self._ws is None or self._ws.numel() < need + 256 or self._ws.device != dev:
            self._ws = torch.empty(int(need) + 256, dtype=torch.uint8, device=dev)
        ws = self._ws[(-self._ws.data_ptr()) % 256:]
        p = lambda t: C.c_void_p(t.data_ptr())
        pm = self.prune_min_kpts if prune_min_kpts is None else prune_min_kpts
        _lib.check(lib.xfh_lg_match_pairs(h, p(kpts), p(desc), p(counts), P, cap, float(size[0]), float(size[1]), float(min_conf), int(pm),
                                          p(matches), p(scores), p(n), p(ws), need,
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream)), "xfh_lg_match_pairs")
        return matches, scores, n

    @torch.inference_mode()
    def forward(self, data, min_conf=0.1):
        """data: keypoints0/1 (1,N,2), descriptors0/1 (1,N,64), image_size0/1 (1,2) as in the reference.
        Returns kornia's result keys: 'matches' (list of (S,2) int64), 'scores' (list of (S,)) -- what
        XFeat.match_lighterglue reads -- plus 'matches0' (1,M) / 'matches1' (1,N) with -1 for unmatched points,
        'matching_scores0/1' (score of the accepted match, 0 elsewhere; kornia also reports mutual-but-below-threshold
        scores there) and 'stop' (= n_layers: early stopping is disabled by the reference's configuration)."""
        self.conf["filter_threshold"] = min_conf
        if data['keypoints0'].shape[0] != 1:
            raise ValueError("LighterGlue supports one pair per call (B = 1), like the reference")
        sz = lambda t: [float(v) for v in torch.as_tensor(t).reshape(-1)[:2].tolist()]
        m, s, c = self.match_device(data['keypoints0'][0], data['descriptors0'][0], sz(data['image_size0']), data['keypoints1'][0],
                                    data['descriptors1'][0], sz(data['image_size1']), min_conf)
        n = int(c.item())
        m, s = m[:n], s[:n]
        n0, n1 = data['keypoints0'].shape[1], data['keypoints1'].shape[1]
        m0 = torch.full((1, n0), -1, dtype=torch.int64, device=m.device)
        m1 = torch.full((1, n1), -1, dtype=torch.int64, device=m.device)
        s0 = torch.zeros((1, n0), dtype=torch.float32, device=m.device)
        s1 = torch.zeros((1, n1), dtype=torch.float32, device=m.device)
        m0[0, m[:, 0]], m1[0, m[:, 1]] = m[:, 1], m[:, 0]
        s0[0, m[:, 0]], s1[0, m[:, 1]] = s, s
        return {'matches0': m0, 'matches1': m1, 'matching_scores0': s0, 'matching_scores1': s1, 'stop': self.conf["n_layers"],
                'matches': [m], 'scores': [s]}
