"""MI355X-native XFeat inference, drop-in for the reference's ``modules/xfeat.py::XFeat``.

Same class name, method names, argument names/defaults and return types as the reference
(/root/reference/modules/xfeat.py:17-403, hubconf.py:5-15), so existing callers only change
the import.  All arithmetic runs in hand-written HIP kernels for gfx950 reached through the
C ABI of ``libxfeat_hip.so`` (include/xfeat_hip.h); PyTorch is used for device memory, streams
and the tensors handed back to the caller.  There is no second backend: without a GPU, or
without the shared library, every inference call raises.
"""
import ctypes as C
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .spec import CONVS, FINE

# the library's defaults of the options the range fallback below has to know (csrc/kernels.hpp: Options; tests/test_gpu_parity.py checks that the mirrors agree)
FX_CONV64, FX_CONV24, FX_HEADS, FX_FINE = 1, 2, 8, 2048      # include/xfeat_hip.h: XFH_FX_*
DEFAULT_FX = FX_CONV64 | FX_CONV24 | FX_HEADS | FX_FINE
DEFAULT_BLOCK1 = 7

__all__ = ["XFeat", "XFeatModel"]


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _Group(nn.Module):
    """Bare container so parameter names match the reference's nested Sequential modules."""


def _attach(root, dotted, tensor, buffer):
    mod = root
    parts = dotted.split(".")
    for p in parts[:-1]:
        if not hasattr(mod, p):
            mod.add_module(p, _Group())
        mod = getattr(mod, p)
    if buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


class _FineMatcher(_Group):
    """``net.fine_matcher(x)``: (n,128) -> (n,64)   (reference: modules/model.py:97-111)."""

    def forward(self, x):
        return self._owner()._fine_matcher(x)


class _U8Image:
    """uint8 pixels on the device plus the divisor the reference applies on the host (255 in parse_input, 1 for .float()).
    Internal: lets match_xfeat / match_xfeat_star keep numpy images as bytes all the way into xfh_backbone_u8."""

    def __init__(self, data, divisor):
        self.data, self.divisor = data, divisor

    @property
    def shape(self):
        return self.data.shape


class _LazyResize:
    """A bilinear resize that has not happened yet: `data` (B,C,Hin,Win) fp32 to be interpolated to (Hm,Wm) with source
    steps (s1h,s1w), optionally followed by preprocess_tensor's resize to (Ho,Wo).  Internal: lets extract_dualscale hand
    both resizes to xfh_backbone_resized instead of materialising two images per scale."""

    def __init__(self, data, Hm, Wm, s1h, s1w, Ho=None, Wo=None, s2h=1.0, s2w=1.0):
        self.data, self.mid, self.s1 = data, (Hm, Wm), (s1h, s1w)
        self.out, self.s2 = (Hm if Ho is None else Ho, Wm if Wo is None else Wo), (s2h, s2w)

    @property
    def shape(self):
        return (self.data.shape[0], self.data.shape[1]) + tuple(self.out)

    @property
    def device(self):
        return self.data.device


class XFeatModel(nn.Module):
    """Parameter container with the reference's ``state_dict`` keys (modules/model.py:33-111);
    ``forward`` runs the HIP backbone and returns the same triple as the reference:
    feats (B,64,H/8,W/8), keypoint logits (B,65,H/8,W/8), reliability (B,1,H/8,W/8).
    feats and logits are channels-last in memory (same shape/values, different strides)."""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        for c in CONVS:
            fan_in = c.cin * c.k * c.k
            bound = 1.0 / math.sqrt(fan_in)
            w = (torch.rand((c.cout, c.cin, c.k, c.k), generator=g) * 2 - 1) * bound
            if c.kind == "bn":
                _attach(self, f"{c.name}.layer.0.weight", w, False)
                _attach(self, f"{c.name}.layer.1.running_mean", torch.zeros(c.cout), True)
                _attach(self, f"{c.name}.layer.1.running_var", torch.ones(c.cout), True)
                _attach(self, f"{c.name}.layer.1.num_batches_tracked", torch.tensor(0, dtype=torch.long), True)
            else:
                _attach(self, f"{c.name}.weight", w, False)
                _attach(self, f"{c.name}.bias", (torch.rand(c.cout, generator=g) * 2 - 1) * bound, False)
        self.add_module("fine_matcher", _FineMatcher())
        for li, fin, fout, bi in FINE:
            bound = 1.0 / math.sqrt(fin)
            _attach(self, f"fine_matcher.{li}.weight", (torch.rand((fout, fin), generator=g) * 2 - 1) * bound, False)
            _attach(self, f"fine_matcher.{li}.bias", (torch.rand(fout, generator=g) * 2 - 1) * bound, False)
            if bi is not None:
                _attach(self, f"fine_matcher.{bi}.running_mean", torch.zeros(fout), True)
                _attach(self, f"fine_matcher.{bi}.running_var", torch.ones(fout), True)
                _attach(self, f"fine_matcher.{bi}.num_batches_tracked", torch.tensor(0, dtype=torch.long), True)
        me = self
        self.fine_matcher._owner = lambda: me          # not a registered submodule cycle
        self._handle = None
        self._handle_device = None
        self._options = {}
        self._ws = {}
        self._ws_epoch = 0

    # -- weights -> C handle -------------------------------------------------------------------
    def weight_arrays(self):
        """The fp32 host arrays in the canonical order of include/xfeat_hip.h (xfh_create)."""
        sd = self.state_dict()
        out = []
        for c in CONVS:
            if c.kind == "bn":
                keys = [f"{c.name}.layer.0.weight", f"{c.name}.layer.1.running_mean", f"{c.name}.layer.1.running_var"]
            else:
                keys = [f"{c.name}.weight", f"{c.name}.bias"]
            out += [sd[k] for k in keys]
        for li, fin, fout, bi in FINE:
            out += [sd[f"fine_matcher.{li}.weight"], sd[f"fine_matcher.{li}.bias"]]
            if bi is not None:
                out += [sd[f"fine_matcher.{bi}.running_mean"], sd[f"fine_matcher.{bi}.running_var"]]
        return [np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy()) for t in out]

    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self._drop_handle()
        return r

    def _load_from_state_dict(self, *args, **kw):
        # reached when a PARENT module (XFeat.load_state_dict) loads weights: the packed device copy is stale.
        # (In-place edits of a parameter tensor are not tracked: call load_state_dict, or net._drop_handle().)
        self._drop_handle()
        return super()._load_from_state_dict(*args, **kw)

    def _drop_handle(self):
        if getattr(self, "_handle", None):
            _lib.load().xfh_destroy(self._handle)
            self._ws_epoch = getattr(self, "_ws_epoch", 0) + 1
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
        arrs = self.weight_arrays()
        n = lib.xfh_num_weight_arrays()
        if len(arrs) != n:
            raise _lib.XFeatHipError(f"weight table has {len(arrs)} arrays, library expects {n}")
        for i, a in enumerate(arrs):
            if a.size != lib.xfh_weight_array_floats(i):
                raise _lib.XFeatHipError(f"weight array {i} has {a.size} floats, expected {lib.xfh_weight_array_floats(i)}")
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        h = C.c_void_p()
        _lib.check(lib.xfh_create(ptrs, n, dev, C.byref(h)), "xfh_create")
        try:                                           # every remembered option applies, or the handle is never published
            for k, v in self._options.items():
                _lib.check(lib.xfh_set_option(h, k.encode(), int(v)), f"xfh_set_option({k})")
        except Exception:
            lib.xfh_destroy(h)
            raise
        self._handle, self._handle_device = h, dev
        self._status = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", dev))      # the handle's status word (include/xfeat_hip.h: xfh_set_status_buffer)
        self._status_target = self._status
        _lib.check(lib.xfh_set_status_buffer(h, C.c_void_p(self._status.data_ptr())), "xfh_set_status_buffer")
        return h

    # include/xfeat_hip.h: xfh_set_option (= api.hip: xfh_set_option's checks)
    OPTION_VALUES = {"match_exact": lambda v: v in (0, 1), "match_sweep": lambda v: v in (0, 1, 2), "resize2": lambda v: v in (0, 1), "block1": lambda v: v in (5, 7), "fx": lambda v: v >= 0 and (v & ~DEFAULT_FX) == 0}

    def set_option(self, key, value):
        """Kernel switch of this model's handle (include/xfeat_hip.h: xfh_set_option): the default kernel of a layer family or its fp32-range fallback.
        Remembered across handle re-creation (load_state_dict, device change).  value None restores the default."""
        if value is None:
            if self._options.pop(key, None) is not None:
                self._drop_handle()                    # a fresh handle starts from the defaults
            return
        # validated BEFORE it is remembered (a bad entry in _options would fail every later handle creation half-way through the list): against the
        # live handle when there is one, against the table below (= api.hip: option_slot) otherwise
        ok = self.OPTION_VALUES.get(key)
        if ok is None or not ok(int(value)):
            raise _lib.XFeatHipError(f"set_option: unknown option or value: {key} = {value} (options: {sorted(self.OPTION_VALUES)}; include/xfeat_hip.h)")
        if self._handle is not None:
            _lib.check(_lib.load().xfh_set_option(self._handle, key.encode(), int(value)), f"xfh_set_option({key})")
        self._options[key] = int(value)

    def set_status_target(self, t=None):
        """The device int32 the handle's kernels report into (bit 0: an activation outside the range of the fp16-pair arithmetic, option `fx`): `t` (a caller's
        tensor, e.g. a slot of the buffer it reads back anyway -- FrameStream) or None for the model's own word.  The caller zeroes / reads its own target."""
        h = self.handle()
        self._status_target = self._status if t is None else t
        assert self._status_target.dtype == torch.int32 and self._status_target.is_cuda
        _lib.check(_lib.load().xfh_set_status_buffer(h, C.c_void_p(self._status_target.data_ptr())), "xfh_set_status_buffer")

    def status_into(self, t):
        """Context manager: while it is open the handle's kernels report into `t` (a zeroed int32 device slot of a buffer the caller reads back anyway), afterwards
        into the model's own word again -- the status costs no read-back of its own.  Kernels take the pointer when they are enqueued, so closing the context
        right behind the last launch is safe."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            prev = getattr(self, "_status_target", None)       # a caller's own target (FrameStream lane, tracker) comes back when the context closes
            self.set_status_target(t)
            try:
                yield t
            finally:
                if self._handle is not None:
                    self.set_status_target(None if prev is None or prev is self._status else prev)
        return cm()

    def clear_status(self):
        """Zero the model's own status word on the current stream (no read-back): a path that does not read the word back itself starts from a clean one."""
        if self._handle is not None and getattr(self, "_status", None) is not None:
            self._status.zero_()

    def take_status(self):
        """Read and clear the model's own status word (one 4-byte read-back; 0 without a handle)."""
        if self._handle is None or getattr(self, "_status_target", None) is None:
            return 0
        v = int(self._status_target[0].item())
        if v:
            self._status_target[:1].zero_()
        return v

    def _effective_option(self, key):
        return self._options.get(key, {"fx": DEFAULT_FX, "block1": DEFAULT_BLOCK1}[key])

    def fx_range_exceeded(self, status=None):
        """True if a call since the last check left the range of the fp16-pair arithmetic (|activation| >= 65504; never seen on images): the model then falls
        back for good to the kernels with fp32's range -- the f32-MFMA convolutions, heads and linear layers (option fx = 0) and the vector-ALU block1 (block1 = 5;
        include/xfeat_hip.h: THE RANGE FALLBACK) -- and the caller repeats the call; results are exact either way.  Written against the EFFECTIVE options (override or
        library default), so it holds whatever the defaults are."""
        fx_on = self._effective_option("fx") != 0 or self._effective_option("block1") == 7
        if status is None and not fx_on:
            return False                               # (every kernel has fp32's range: nothing to read back)
        v = self.take_status() if status is None else int(status)
        if not (v & 1):
            return False
        if not fx_on:
            return False                               # (a stale flag of a caller's buffer: nothing left to switch off, and repeating the call would not end)
        import warnings
        warnings.warn("accelerated_features_amd: an activation left the range of the fp16-pair arithmetic (|x| >= 65504); this model falls back to the "
                      "f32-MFMA / vector-ALU kernels (options fx = 0, block1 = 5) and the call is repeated")
        self.set_option("fx", 0)
        self.set_option("block1", 5)
        return True

    def workspace(self, name, nbytes):
        dev = torch.device("cuda", torch.cuda.current_device())
        t = self._ws.get(name)
        if t is None or t.device != dev or t.numel() - ((-t.data_ptr()) % 256) < nbytes:
            t = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=dev)
            self._ws[name] = t
            self._ws_epoch += 1                # hipGraphs captured on the old buffer are stale (graphs.py checks this)
        off = (-t.data_ptr()) % 256
        return t[off:], int(nbytes)

    # -- the network -----------------------------------------------------------------------------
    def backbone(self, x, want_logits=True, want_heat=False, want_invnorm=False):
        """x (B,C,H,W) float32 CUDA, H%32==W%32==0 -> (feats_cl (B,h,w,64), logits_cl|None, heat|None, rel (B,h,w)); with
        want_invnorm a fifth element: 1/max(||feats[b,i,j,:]||, 1e-12) (B,h,w), the reliability head's by-product that
        xfh_detect_sparse would otherwise recompute (F.normalize(M1, dim=1), xfeat.py:70)."""
        if len(x.shape) != 4:
            raise RuntimeError("Input tensor needs to be in (B,C,H,W) format")
        lib = _lib.load()
        h = self.handle()
        u8_div = None
        probe = x.data if isinstance(x, (_LazyResize, _U8Image)) else x
        if not probe.is_cuda:
            raise _lib.XFeatHipError("XFeatModel runs on the GPU: got a CPU tensor (move the input with .cuda(); the XFeat.* "
                                     "methods do that themselves)")
        if isinstance(x, _LazyResize):
            B, Cc, H, W = x.shape
            hc, wc = H // 8, W // 8
            dev = x.device
            feats = torch.empty((B, hc, wc, 64), dtype=torch.float32, device=dev)
            rel = torch.empty((B, hc, wc), dtype=torch.float32, device=dev)
            logits = torch.empty((B, hc, wc, 65), dtype=torch.float32, device=dev) if want_logits else None
            heat = torch.empty((B, H, W), dtype=torch.float32, device=dev) if want_heat else None
            inv = torch.empty((B, hc, wc), dtype=torch.float32, device=dev) if want_invnorm else None
            ws, n = self.workspace("backbone", lib.xfh_backbone_workspace_bytes(B, Cc, H, W))
            _lib.check(lib.xfh_backbone_resized(h, _ptr(x.data), B, Cc, x.data.shape[2], x.data.shape[3], x.mid[0], x.mid[1],
                                                float(x.s1[0]), float(x.s1[1]), H, W, float(x.s2[0]), float(x.s2[1]), _ptr(feats),
                                                _ptr(logits), _ptr(heat), _ptr(rel), _ptr(inv), _ptr(ws), n, _stream()), "xfh_backbone_resized")
            return (feats, logits, heat, rel, inv) if want_invnorm else (feats, logits, heat, rel)
        if isinstance(x, _U8Image):
            x, u8_div = x.data, x.divisor
        elif x.dtype == torch.uint8:
            u8_div = 1.0                    # reference: x.float() (modules/xfeat.py:232), no scaling
        if u8_div is None:
            x = x.contiguous().float()
        B, Cc, H, W = x.shape
        hc, wc = H // 8, W // 8
        dev = x.device
        feats = torch.empty((B, hc, wc, 64), dtype=torch.float32, device=dev)
        rel = torch.empty((B, hc, wc), dtype=torch.float32, device=dev)
        logits = torch.empty((B, hc, wc, 65), dtype=torch.float32, device=dev) if want_logits else None
        heat = torch.empty((B, H, W), dtype=torch.float32, device=dev) if want_heat else None
        inv = torch.empty((B, hc, wc), dtype=torch.float32, device=dev) if want_invnorm else None
        ret = (feats, logits, heat, rel, inv) if want_invnorm else (feats, logits, heat, rel)
        ws, n = self.workspace("backbone", lib.xfh_backbone_workspace_bytes(B, Cc, H, W))
        if u8_div is not None:
            # uint8 pixels go to the device as they are (a quarter of the bytes); numpy HWC images arrive as a permuted
            # view whose memory is (B,H,W,C): no host-side transpose either
            if x.permute(0, 2, 3, 1).is_contiguous() and not x.is_contiguous():
                layout, xb = _lib.LAYOUT_NHWC, x
            else:
                layout, xb = _lib.LAYOUT_NCHW, x.contiguous()
            _lib.check(lib.xfh_backbone_u8(h, _ptr(xb), layout, float(u8_div), B, Cc, H, W, _ptr(feats), _ptr(logits), _ptr(heat),
                                           _ptr(rel), _ptr(inv), _ptr(ws), n, _stream()), "xfh_backbone_u8")
            return ret
        _lib.check(lib.xfh_backbone(h, _ptr(x), B, Cc, H, W, _ptr(feats), _ptr(logits), _ptr(heat), _ptr(rel), _ptr(inv),
                                    _ptr(ws), n, _stream()), "xfh_backbone")
        return ret

    def forward(self, x):
        """feats, keypoint logits, reliability like the reference's XFeatModel.forward; everything stays on the device and nothing is read back: the status word of the
        fp16-pair arithmetic (|activation| >= 65504: never seen on images) is the caller's to check -- `fx_range_exceeded()` after the call, then repeat (XFeat's
        detectAndCompute / match_xfeat* do it inside their own read-backs)."""
        feats, logits, _, rel = self.backbone(x, want_logits=True, want_heat=False)
        return feats.permute(0, 3, 1, 2), logits.permute(0, 3, 1, 2), rel[:, None]

    def _fine_matcher(self, x):
        lib = _lib.load()
        h = self.handle()
        if not x.is_cuda:
            raise _lib.XFeatHipError("fine_matcher runs on the GPU: got a CPU tensor")
        x = x.contiguous().float()
        n = x.shape[0]
        out = torch.empty((n, 64), dtype=torch.float32, device=x.device)
        if n == 0:
            return out
        ws, nb = self.workspace("refine", lib.xfh_refine_workspace_bytes(1, n))
        _lib.check(lib.xfh_fine_matcher(h, _ptr(x), n, _ptr(out), _ptr(ws), nb, _stream()), "xfh_fine_matcher")
        return out


def _counts_pair(n_valid, n_cand):
    """(2,B) tensor of the two count rows: the buffer they are views of (as _detect_call allocates them) or a stacked copy."""
    base = getattr(n_valid, "_base", None)
    if base is not None and base is getattr(n_cand, "_base", None) and base.dim() == 2 and base.shape[0] >= 2 and base.is_contiguous() \
            and n_valid.data_ptr() == base.data_ptr() and n_cand.data_ptr() == base[1].data_ptr():
        return base[:2]
    return torch.stack([n_valid, n_cand])


class XFeat(nn.Module):
    """
        Implements the inference module for XFeat (sparse and semi-dense extraction & matching)
        on MI355X.  Mirrors /root/reference/modules/xfeat.py::XFeat.
    """

    def __init__(self, weights=os.path.abspath(os.path.dirname(__file__)) + '/../weights/xfeat.pt', top_k=4096,
                 detection_threshold=0.05):
        super().__init__()
        self.dev = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        self.net = XFeatModel().eval()
        self.top_k = top_k
        self.detection_threshold = detection_threshold

        if weights is not None:
            if isinstance(weights, str):
                print('loading weights from: ' + weights)
                self.net.load_state_dict(torch.load(weights, map_location='cpu'))
            else:
                self.net.load_state_dict(weights)

        from .interpolator import InterpolateSparse2d
        self.interpolator = InterpolateSparse2d('bicubic')      # reference attribute (xfeat.py:37); the hot path fuses its sampling
        # kornia probe like the reference (xfeat.py:39-46): informational here -- match_lighterglue runs on the HIP matcher
        self.kornia_available = False
        self.lighterglue = None
        try:
            import importlib.util
            self.kornia_available = importlib.util.find_spec("kornia") is not None
        except Exception:
            pass

    def set_option(self, key, value):
        """Kernel-variant switch (xfh_set_option; see XFeatModel.set_option)."""
        self.net.set_option(key, value)

    # ------------------------------------------------------------------------------------------
    # sparse
    # ------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def detectAndCompute(self, x, top_k=None, detection_threshold=None):
        """
            Compute sparse keypoints & descriptors. Supports batched mode.

            input:
                x -> torch.Tensor(B, C, H, W): grayscale or rgb image
                top_k -> int: keep best k features
            return:
                List[Dict]:
                    'keypoints'    ->   torch.Tensor(N, 2): keypoints (x,y)
                    'scores'       ->   torch.Tensor(N,): keypoint scores
                    'descriptors'  ->   torch.Tensor(N, 64): local features
        """
        cap = None
        B = x.shape[0] if len(getattr(x, "shape", ())) == 4 else 1      # (a numpy (H,W[,C]) image is one frame; other shapes are refused by preprocess_tensor)
        while True:
            self._require_gpu()
            cnt_dev = torch.zeros((3, B), dtype=torch.int32, device=self.dev)      # n_valid, n_candidates, [2, 0] = the status word of the fp16-pair arithmetic
            with self.net.status_into(cnt_dev[2]):
                kpts, scores, desc, n_valid, n_cand, cap, hw = self._detect_device(x, top_k, detection_threshold, cap, counts_out=cnt_dev[:2])
            cnt = cnt_dev.cpu()                                  # the ONE read-back per batch (counts and status together)
            if self.net.fx_range_exceeded(status=int(cnt[2, 0])):      # (fp16-pair arithmetic out of range: never on images; exact re-run on the fp32-range kernels)
                continue
            ncmax = int(cnt[1].max())
            if cap >= hw or ncmax <= cap:
                break
            cap = min(hw, max(ncmax, 2 * cap))                   # plateau image: exact re-run with room
        nv = cnt[0].tolist()
        K = kpts.shape[1]
        if min(nv) == K:            # every image filled its top_k (the usual case): three unbind calls instead of 3 B slicing calls (0.2 ms of Python per 64-frame batch)
            return [{'keypoints': k, 'scores': s, 'descriptors': d} for k, s, d in zip(kpts.unbind(0), scores.unbind(0), desc.unbind(0))]
        return [{'keypoints': kpts[b, :nv[b]], 'scores': scores[b, :nv[b]], 'descriptors': desc[b, :nv[b]]}
                for b in range(len(nv))]

    @torch.inference_mode()
    def detectAndComputePadded(self, x, top_k=None, detection_threshold=None, counts_out=None, with_f16=True):
        """detectAndCompute without the ragged lists and without a read-back: the throughput form of the same call (everything stays on the device; pair it with
        match_pairs_device, or use streaming.FrameStream / batching.match_pairs, which do).  Returns a dict of fixed-capacity tensors
            'keypoints' (B,top_k,2), 'scores' (B,top_k), 'descriptors' (B,top_k,64), 'n_valid' (B,) int32  -- rows [n_valid[b]:] of image b are unspecified,
            'n_candidates' (B,) int32 and 'nms_capacity': if n_candidates.max() > nms_capacity a plateau image overflowed the candidate list and the
            call has to be repeated through detectAndCompute (which does so by itself); never seen on natural or textured images,
            'descriptors_f16' (with_f16): 256 x the descriptors rounded to fp16, the copy match_pairs_device's filter reads.
        counts_out: an int32 (2,B) device tensor to receive n_valid / n_candidates (a caller that keeps all its counts in one buffer reads them back with one copy).
        The status word of the fp16-pair arithmetic (XFeatModel.fx_range_exceeded) is the caller's to check."""
        out = self._detect_device(x, top_k, detection_threshold, None, with_f16, counts_out)
        d = {'keypoints': out[0], 'scores': out[1], 'descriptors': out[2], 'n_valid': out[3], 'n_candidates': out[4], 'nms_capacity': out[5]}
        if with_f16:
            d['descriptors_f16'] = out[7]
        return d

    def _detect_device(self, x, top_k=None, detection_threshold=None, cap=None, want_f16=False, counts_out=None):
        """Fixed-capacity device results, no read-back: kpts (B,top_k,2), scores (B,top_k),
        desc (B,top_k,64), n_valid (B) int32, n_cand (B) int32, the NMS capacity used, H*W
        (+ with want_f16 an eighth element: 256 * descriptors rounded to fp16, (B,top_k,64) float16, for match_pairs_device).
        If n_cand.max() > capacity the candidate list was truncated (caller re-runs).
        n_valid / n_cand are the two rows of ONE (2,B) tensor (counts_out if given: a caller that also matches can keep all its counts
        in one buffer and read them back with one copy, without a concatenation kernel)."""
        if top_k is None: top_k = self.top_k
        if detection_threshold is None: detection_threshold = self.detection_threshold
        x, rh1, rw1 = self.preprocess_tensor(x)
        B, _, H, W = x.shape
        feats, _, heat, rel, inv = self.net.backbone(x, want_logits=False, want_heat=True, want_invnorm=True)
        if cap is None:
            cap = min(H * W, max(int(top_k), (H * W) // 8))
        out = self._detect_call(feats, heat, rel, B, H, W, detection_threshold, int(top_k), cap, rw1, rh1, inv, want_f16, counts_out)
        if want_f16:
            return out[0], out[1], out[2], out[3], out[4], cap, H * W, out[5]
        return out[0], out[1], out[2], out[3], out[4], cap, H * W

    def _detect_call(self, feats, heat, rel, B, H, W, thr, top_k, cap, rw, rh, inv=None, want_f16=False, counts_out=None):
        lib = _lib.load()
        dev = feats.device
        kpts = torch.empty((B, top_k, 2), dtype=torch.float32, device=dev)
        scores = torch.empty((B, top_k), dtype=torch.float32, device=dev)
        desc = torch.empty((B, top_k, 64), dtype=torch.float32, device=dev)
        cnt = counts_out if counts_out is not None else torch.empty((2, B), dtype=torch.int32, device=dev)
        assert cnt.shape == (2, B) and cnt.dtype == torch.int32 and cnt[0].is_contiguous() and cnt[1].is_contiguous()
        n_valid, n_cand = cnt[0], cnt[1]
        d16 = torch.empty((B, top_k, 64), dtype=torch.float16, device=dev) if want_f16 else None
        ws, n = self.net.workspace("detect", lib.xfh_detect_workspace_bytes(B, H, W, top_k, cap))
        _lib.check(lib.xfh_detect_sparse(self.net.handle(), _ptr(heat), _ptr(rel), _ptr(feats), _ptr(inv), B, H, W, float(thr), top_k, cap,
                                         float(rw), float(rh), _ptr(kpts), _ptr(scores), _ptr(desc), _ptr(d16), _ptr(n_valid),
                                         _ptr(n_cand), _ptr(ws), n, _stream()), "xfh_detect_sparse")
        return kpts, scores, desc, n_valid, n_cand, d16

    # ------------------------------------------------------------------------------------------
    # semi-dense
    # ------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def detectAndComputeDense(self, x, top_k=None, multiscale=True):
        """
            Compute dense *and coarse* descriptors. Supports batched mode.
            return: features sorted by their reliability score -- from most to least
                Dict:
                    'keypoints'    ->   torch.Tensor(B, top_k, 2): coarse keypoints
                    'scales'       ->   torch.Tensor(B, top_k): extraction scale
                    'descriptors'  ->   torch.Tensor(B, top_k, 64): coarse local features
        """
        # The reference's dict, nothing added.  No read-back here (the results stay on the device, as in the reference), so the range guard of the fp16-pair arithmetic
        # (never seen to fire on images) is the caller's to check: `self.net.fx_range_exceeded()` after the call reads the model's status word (4 bytes), switches the
        # model to the fp32-range kernels if it was set and returns True -- then repeat the call.  match_xfeat_star does exactly that.  The word is cleared before the
        # kernels are enqueued, so a flag left by an earlier call cannot trigger a spurious fallback.
        self.net.clear_status()
        return self._dense_device(x, top_k, multiscale)

    def _dense_device(self, x, top_k=None, multiscale=True):
        """detectAndComputeDense without the status read-back (match_xfeat_star checks once per call)."""
        if top_k is None: top_k = self.top_k
        if multiscale:
            mkpts, sc, feats = self.extract_dualscale(x, top_k)
        else:
            mkpts, feats = self.extractDense(x, top_k)
            sc = torch.ones(mkpts.shape[:2], device=mkpts.device)
        return {'keypoints': mkpts, 'descriptors': feats, 'scales': sc}

    @torch.inference_mode()
    def match_lighterglue(self, d0, d1, min_conf=0.1):
        """
            Match XFeat sparse features with LightGlue (smaller version) -- one pair per call, like the reference
            (/root/reference/modules/xfeat.py:131-162).
            input:
                d0, d1: Dict('keypoints', 'scores, 'descriptors', 'image_size (Width, Height)')
            output:
                mkpts_0, mkpts_1 -> np.ndarray (N,2) xy coordinate matches from image1 to image2
                idx              -> np.ndarray (N,2) the indices of the matching features
        """
        self._require_gpu()
        if self.lighterglue is None:
            from .lighterglue import LighterGlue
            self.lighterglue = LighterGlue()
        m, _, c = self.lighterglue.match_device(d0['keypoints'], d0['descriptors'], d0['image_size'], d1['keypoints'],
                                                d1['descriptors'], d1['image_size'], min_conf)
        idxs = m[:int(c.item())]
        k0, k1 = d0['keypoints'].to(idxs.device), d1['keypoints'].to(idxs.device)
        return k0[idxs[:, 0]].cpu().numpy(), k1[idxs[:, 1]].cpu().numpy(), idxs.cpu().numpy()

    @torch.inference_mode()
    def match_xfeat(self, img1, img2, top_k=None, min_cossim=-1):
        """
            Simple extractor and MNN matcher (B=1).
            returns:
                mkpts_0, mkpts_1 -> np.ndarray (N,2) xy coordinate matches from image1 to image2
        """
        if top_k is None: top_k = self.top_k
        img1 = self._parse_input_fast(img1)
        img2 = self._parse_input_fast(img2)

        out1 = self.detectAndCompute(img1, top_k=top_k)[0]
        out2 = self.detectAndCompute(img2, top_k=top_k)[0]

        idxs0, idxs1 = self.match(out1['descriptors'], out2['descriptors'], min_cossim=min_cossim)

        return out1['keypoints'][idxs0].cpu().numpy(), out2['keypoints'][idxs1].cpu().numpy()

    @torch.inference_mode()
    def match_xfeat_star(self, im_set1, im_set2, top_k=None):
        """
            Extracts coarse feats, then match pairs and finally refine matches (batched).
            returns:
                matches -> List[torch.Tensor(N, 4)]: List of size B of pairwise matches (x1,y1,x2,y2);
                           for B == 1 a tuple of two numpy (N,2) arrays, like the reference.
        """
        if top_k is None: top_k = self.top_k
        im_set1 = self.parse_input(im_set1)          # the dual-scale path resizes first: it needs the float image
        im_set2 = self.parse_input(im_set2)

        while True:
            out1 = self._dense_device(im_set1, top_k=top_k)
            out2 = self._dense_device(im_set2, top_k=top_k)

            idx0, idx1, n_matches = self._batch_match_device(out1['descriptors'], out2['descriptors'], -1)
            out, n_out = self._refine_device(out1, out2, idx0, idx1, n_matches, 0.25)
            counts = torch.cat([n_out.to(torch.int32), self.net._status_target[:1]]).cpu().tolist()      # the ONE read-back: the counts and the status word behind them
            status = counts.pop()
            if status:
                self.net._status_target[:1].zero_()
            if not self.net.fx_range_exceeded(status=status):        # (see detectAndCompute)
                break
        matches = [out[b, :counts[b]] for b in range(len(counts))]
        B = len(im_set1)
        return matches if B > 1 else (matches[0][:, :2].cpu().numpy(), matches[0][:, 2:].cpu().numpy())

    # ------------------------------------------------------------------------------------------
    # pre-processing
    # ------------------------------------------------------------------------------------------
    def preprocess_tensor(self, x):
        """ Guarantee that image is divisible by 32 to avoid aliasing artifacts. """
        if isinstance(x, np.ndarray):
            if len(x.shape) == 3:
                x = torch.tensor(x).permute(2, 0, 1)[None]
            elif len(x.shape) == 2:
                x = torch.tensor(x[..., None]).permute(2, 0, 1)[None]
            else:
                raise RuntimeError('For numpy arrays, only (H,W) or (H,W,C) format is supported.')

        if len(x.shape) != 4:
            raise RuntimeError('Input tensor needs to be in (B,C,H,W) format')

        self._require_gpu()
        H, W = x.shape[-2:]
        _H, _W = (H // 32) * 32, (W // 32) * 32
        if _H == 0 or _W == 0:
            raise RuntimeError('Input image must be at least 32x32 pixels')
        rh, rw = H / _H, W / _W
        if isinstance(x, _LazyResize):
            s2h, s2w = np.float32(H) / np.float32(_H), np.float32(W) / np.float32(_W)
            if max(s2h, s2w) < 1.9 and x.data.shape[1] <= 4 and x.data[0].numel() * 4 < 2 ** 31:
                return _LazyResize(x.data, x.mid[0], x.mid[1], x.s1[0], x.s1[1], _H, _W, s2h, s2w), rh, rw
            x = self._resize(x.data, x.mid[0], x.mid[1], x.s1[0], x.s1[1])      # tiny images: materialise
        if (_H, _W) == (H, W) and (isinstance(x, _U8Image) or x.dtype == torch.uint8):
            # bytes stay bytes: the conversion (and parse_input's /255) happens inside xfh_backbone_u8
            if isinstance(x, _U8Image):
                return _U8Image(x.data.to(self.dev), x.divisor), rh, rw
            return x.to(self.dev), rh, rw
        if isinstance(x, _U8Image):
            x = x.data / x.divisor                              # resize path: convert like the reference, then interpolate
        x = x.to(self.dev).float().contiguous()
        if (_H, _W) != (H, W):
            x = self._resize(x, _H, _W, np.float32(H) / np.float32(_H), np.float32(W) / np.float32(_W))
        return x, rh, rw

    def _resize(self, x, Hout, Wout, scale_h, scale_w):
        """bilinear, align_corners=False, with the source step PyTorch would use."""
        lib = _lib.load()
        B, Cc, H, W = x.shape
        out = torch.empty((B, Cc, Hout, Wout), dtype=torch.float32, device=x.device)
        _lib.check(lib.xfh_resize_bilinear(_ptr(x), B * Cc, H, W, _ptr(out), Hout, Wout, float(scale_h), float(scale_w),
                                           _stream()), "xfh_resize_bilinear")
        return out

    def _require_gpu(self):
        if self.dev.type != 'cuda':
            raise _lib.XFeatHipError("accelerated_features_amd needs an AMD MI355X (gfx950) GPU; no CPU fallback exists")

    # ------------------------------------------------------------------------------------------
    # helper methods of the reference surface
    # ------------------------------------------------------------------------------------------
    def get_kpts_heatmap(self, kpts, softmax_temp=1.0):
        """kpts (B,65,h,w) logits -> (B,1,8h,8w) heat map (xfeat.py:242-247)."""
        self._require_gpu()
        if softmax_temp != 1.0:
            kpts = kpts * softmax_temp
        lib = _lib.load()
        B, Cc, h, w = kpts.shape
        if Cc != 65:
            raise RuntimeError('keypoint logits must have 65 channels')
        cl = kpts.to(self.dev).float().permute(0, 2, 3, 1).contiguous()
        heat = torch.empty((B, 1, h * 8, w * 8), dtype=torch.float32, device=cl.device)
        _lib.check(lib.xfh_kpts_heatmap(_ptr(cl), B, h, w, _ptr(heat), _stream()), "xfh_kpts_heatmap")
        return heat

    def NMS(self, x, threshold=0.05, kernel_size=5):
        """x (B,1,H,W) -> (B,Nmax,2) int64 (x,y), zero padded (xfeat.py:249-263)."""
        if kernel_size < 1 or kernel_size % 2 == 0:
            raise RuntimeError('NMS kernel_size must be odd (the reference pads kernel_size//2 on both sides)')
        self._require_gpu()
        lib = _lib.load()
        x = x.to(self.dev).float().contiguous()
        B, _, H, W = x.shape
        cap = min(H * W, max(4096, (H * W) // 8))
        while True:
            xy = torch.empty((B, cap, 2), dtype=torch.int64, device=x.device)
            nc = torch.empty((B,), dtype=torch.int32, device=x.device)
            ws, n = self.net.workspace("detect", lib.xfh_detect_workspace_bytes(B, H, W, 1, cap))
            _lib.check(lib.xfh_nms(self.net.handle(), _ptr(x), B, H, W, float(threshold), int(kernel_size), cap, _ptr(xy), _ptr(nc),
                                   _ptr(ws), n, _stream()), "xfh_nms")
            nmax = int(nc.max().item())
            if nmax <= cap:
                return xy[:, :nmax]
            cap = min(H * W, max(nmax, 2 * cap))

    @torch.inference_mode()
    def batch_match(self, feats1, feats2, min_cossim=-1):
        idx0, idx1, n = self._batch_match_device(feats1, feats2, min_cossim)
        counts = n.cpu().tolist()
        return [(idx0[b, :counts[b]], idx1[b, :counts[b]]) for b in range(len(counts))]

    def _batch_match_device(self, feats1, feats2, min_cossim):
        self._require_gpu()
        lib = _lib.load()
        f1 = feats1.to(self.dev).float().contiguous()
        f2 = feats2.to(self.dev).float().contiguous()
        P, N1, D = f1.shape
        _, N2, _ = f2.shape
        if D != 64 or f2.shape[0] != P or f2.shape[2] != 64:
            raise RuntimeError('descriptors must be (B,N,64)')
        idx0 = torch.empty((P, N1), dtype=torch.int64, device=f1.device)
        idx1 = torch.empty((P, N1), dtype=torch.int64, device=f1.device)
        n = torch.empty((P,), dtype=torch.int32, device=f1.device)
        ws, nb = self.net.workspace("match", lib.xfh_match_workspace_bytes(P, N1, N2))
        _lib.check(lib.xfh_match_mnn(self.net.handle(), _ptr(f1), N1 * 64, _ptr(f2), N2 * 64, None, None, None, None, 0, 0, P, N1, N2,
                                     float(min_cossim), _ptr(idx0), _ptr(idx1), _ptr(n), _ptr(ws), nb, _stream()),
                   "xfh_match_mnn")
        return idx0, idx1, n

    def match_pairs_device(self, desc, n_valid, min_cossim=-1, desc_f16=None, n_out=None):
        """Match consecutive frames (2i, 2i+1) of one detection batch without any read-back.
        desc (B,top_k,64), n_valid (B) int32 as returned by _detect_device; B even.  desc_f16: the fp16 copy the same
        _detect_device(want_f16=True) call returned (saves the matcher's conversion passes; results identical).
        Returns idx0, idx1 (B/2, top_k) int64 and n_matches (B/2) int32 (n_out if given), all on the device."""
        self._require_gpu()
        lib = _lib.load()
        B, K, D = desc.shape
        assert B % 2 == 0 and D == 64 and desc.is_contiguous()
        P = B // 2
        idx0 = torch.empty((P, K), dtype=torch.int64, device=desc.device)
        idx1 = torch.empty((P, K), dtype=torch.int64, device=desc.device)
        n = n_out if n_out is not None else torch.empty((P,), dtype=torch.int32, device=desc.device)
        assert n.shape == (P,) and n.dtype == torch.int32 and n.is_contiguous()
        ws, nb = self.net.workspace("match", lib.xfh_match_workspace_bytes(P, K, K))
        d2 = desc[1]
        b1 = b2 = None
        if desc_f16 is not None:
            assert desc_f16.shape == desc.shape and desc_f16.dtype == torch.float16 and desc_f16.is_contiguous()
            b1, b2 = desc_f16, desc_f16[1]
        _lib.check(lib.xfh_match_mnn(self.net.handle(), _ptr(desc), 2 * K * 64, _ptr(d2), 2 * K * 64, _ptr(b1), _ptr(b2), _ptr(n_valid), _ptr(n_valid),
                                     2, 1, P, K, K, float(min_cossim), _ptr(idx0), _ptr(idx1), _ptr(n), _ptr(ws), nb,
                                     _stream()), "xfh_match_mnn")
        return idx0, idx1, n

    def match_sets_device(self, desc_a, n_a, desc_b, n_b, min_cossim=-1):
        """Pair p = (desc_a[p], desc_b[p]): two detection batches (P,K,64) with their n_valid (P) int32 (images of different
        size go through _detect_device separately).  Same outputs as match_pairs_device, no read-back."""
        self._require_gpu()
        lib = _lib.load()
        P, K, D = desc_a.shape
        assert desc_b.shape == desc_a.shape and D == 64 and desc_a.is_contiguous() and desc_b.is_contiguous()
        idx0 = torch.empty((P, K), dtype=torch.int64, device=desc_a.device)
        idx1 = torch.empty((P, K), dtype=torch.int64, device=desc_a.device)
        n = torch.empty((P,), dtype=torch.int32, device=desc_a.device)
        ws, nb = self.net.workspace("match", lib.xfh_match_workspace_bytes(P, K, K))
        # one count array so that a single (stride, offset) addresses both sides
        nv = torch.cat([n_a.to(torch.int32), n_b.to(torch.int32)]).contiguous()
        _lib.check(lib.xfh_match_mnn(self.net.handle(), _ptr(desc_a), K * 64, _ptr(desc_b), K * 64, None, None, _ptr(nv), _ptr(nv),
                                     1, P, P, K, K, float(min_cossim), _ptr(idx0), _ptr(idx1), _ptr(n), _ptr(ws), nb,
                                     _stream()), "xfh_match_mnn")
        return idx0, idx1, n

    def subpix_softmax2d(self, heatmaps, temp=3):
        """(N,8,8) -> (N,2) expected offset under softmax(temp*heatmaps) (xfeat.py:292-304).
        Helper kept for API compatibility; refine_matches fuses this step in HIP."""
        N, H, W = heatmaps.shape
        p = torch.softmax(temp * heatmaps.reshape(-1, H * W), -1)
        idx = torch.arange(H * W, device=heatmaps.device)
        gx = (idx % W - W // 2).to(p.dtype)
        gy = (idx // W - H // 2).to(p.dtype)
        return torch.stack([(p * gx).sum(1), (p * gy).sum(1)], -1)

    def refine_matches(self, d0, d1, matches, batch_idx, fine_conf=0.25):
        idx0, idx1 = matches[batch_idx]
        sub0 = {k: v[batch_idx:batch_idx + 1] for k, v in d0.items()}
        sub1 = {k: v[batch_idx:batch_idx + 1] for k, v in d1.items()}
        N = sub0['keypoints'].shape[1]
        dev = sub0['keypoints'].device
        n = len(idx0)
        i0 = torch.zeros((1, N), dtype=torch.int64, device=dev)
        i1 = torch.zeros((1, N), dtype=torch.int64, device=dev)
        if n > N:
            raise RuntimeError('more matches than key-points')
        i0[0, :n] = idx0.to(dev)
        i1[0, :n] = idx1.to(dev)
        nm = torch.tensor([n], dtype=torch.int32, device=dev)
        out, n_out = self._refine_device(sub0, sub1, i0, i1, nm, fine_conf)
        return out[0, :int(n_out.item())]

    def _refine_device(self, d0, d1, idx0, idx1, n_matches, fine_conf):
        self._require_gpu()
        lib = _lib.load()
        f0 = d0['descriptors'].float().contiguous()
        f1 = d1['descriptors'].float().contiguous()
        k0 = d0['keypoints'].float().contiguous()
        k1 = d1['keypoints'].float().contiguous()
        s0 = d0['scales'].float().contiguous()
        P = f0.shape[0]
        N = max(f0.shape[1], f1.shape[1])

        def pad(t):            # sets of different size (small images): zero rows up to the common capacity, never indexed
            if t.shape[1] == N:
                return t
            z = torch.zeros((t.shape[0], N - t.shape[1]) + tuple(t.shape[2:]), dtype=t.dtype, device=t.device)
            return torch.cat([t, z], 1).contiguous()
        f0, f1, k0, k1, s0, idx0, idx1 = pad(f0), pad(f1), pad(k0), pad(k1), pad(s0), pad(idx0), pad(idx1)
        out = torch.empty((P, N, 4), dtype=torch.float32, device=f0.device)
        n_out = torch.empty((P,), dtype=torch.int32, device=f0.device)
        ws, nb = self.net.workspace("refine", lib.xfh_refine_workspace_bytes(P, N))
        _lib.check(lib.xfh_refine_matches(self.net.handle(), _ptr(f0), _ptr(f1), _ptr(k0), _ptr(k1), _ptr(s0),
                                          _ptr(idx0.contiguous()), _ptr(idx1.contiguous()), _ptr(n_matches.contiguous()), P, N, float(fine_conf),
                                          _ptr(out), _ptr(n_out), _ptr(ws), nb, _stream()), "xfh_refine_matches")
        return out, n_out

    @torch.inference_mode()
    def match(self, feats1, feats2, min_cossim=0.82):
        if len(feats1) == 0 or len(feats2) == 0:
            e = torch.empty((0,), dtype=torch.int64, device=self.dev)
            return e, e.clone()
        idx0, idx1, n = self._batch_match_device(feats1[None], feats2[None], min_cossim)
        k = int(n.item())
        return idx0[0, :k], idx1[0, :k]

    @torch.inference_mode()
    def match_many(self, feats1, feats2, min_cossim=0.82):
        """The list form of `match` (modules/xfeat.py:327-348 applied to P pairs): pair p = (feats1[p], feats2[p]), each an (N_p, 64) descriptor tensor.
        Returns a list of P tuples (idx0, idx1) -- exactly what `match(feats1[p], feats2[p], min_cossim)` returns for every p -- from ONE launch sequence and ONE
        read-back (the P match counts) instead of P of each: on a 64-frame batch `[xf.match(a, b) for a, b in pairs]` spends its time in 32 host round trips.
            res = xf.detectAndCompute(frames)                                   # List[Dict], as in the reference
            ms = xf.match_many([r['descriptors'] for r in res[0::2]], [r['descriptors'] for r in res[1::2]])
        Descriptor tensors that are row-prefix views of one padded (B, K, 64) tensor at a constant stride -- what detectAndCompute hands out -- are matched in place
        (no copy; the rows are read as they are at the time of the call, so descriptors modified in place since detectAndCompute are matched as modified); anything
        else is padded into one tensor first."""
        P = len(feats1)
        if len(feats2) != P:
            raise RuntimeError('match_many: the two lists must have the same length')
        if P == 0:
            return []
        self._require_gpu()
        lib = _lib.load()
        lens1, lens2 = [int(f.shape[0]) for f in feats1], [int(f.shape[0]) for f in feats2]
        lay = self._strided_layout(feats1, feats2)
        if lay is None:                                          # general case: pad both sides
            N1, N2 = max(max(lens1), 1), max(max(lens2), 1)
            d1 = torch.zeros((P, N1, 64), dtype=torch.float32, device=self.dev)
            d2 = torch.zeros((P, N2, 64), dtype=torch.float32, device=self.dev)
            for p_, (a, b) in enumerate(zip(feats1, feats2)):
                if a.dim() != 2 or b.dim() != 2 or a.shape[-1] != 64 or b.shape[-1] != 64:
                    raise RuntimeError('descriptors must be (N,64)')
                d1[p_, :lens1[p_]] = a
                d2[p_, :lens2[p_]] = b
            p1, ps1, p2, ps2, h1, h2 = d1.data_ptr(), N1 * 64, d2.data_ptr(), N2 * 64, None, None
            keep = (d1, d2)
        else:
            p1, ps1, p2, ps2, h1, h2, keep = lay
            N1, N2 = max(max(lens1), 1), max(max(lens2), 1)
        nv = torch.tensor(lens1 + lens2, dtype=torch.int32).to(self.dev, non_blocking=True)      # one count array, both sides (n_stride 1, offset P)
        idx0 = torch.empty((P, N1), dtype=torch.int64, device=self.dev)
        idx1 = torch.empty((P, N1), dtype=torch.int64, device=self.dev)
        n = torch.empty((P,), dtype=torch.int32, device=self.dev)
        ws, nb = self.net.workspace("match", lib.xfh_match_workspace_bytes(P, N1, N2))
        _lib.check(lib.xfh_match_mnn(self.net.handle(), C.c_void_p(p1), ps1, C.c_void_p(p2), ps2, C.c_void_p(h1) if h1 else None, C.c_void_p(h2) if h2 else None,
                                     _ptr(nv), _ptr(nv), 1, P, P, N1, N2, float(min_cossim), _ptr(idx0), _ptr(idx1), _ptr(n), _ptr(ws), nb, _stream()), "xfh_match_mnn")
        cnt = n.cpu().tolist()                                   # the one read-back
        del keep
        return [(idx0[p_, :cnt[p_]], idx1[p_, :cnt[p_]]) if lens1[p_] and lens2[p_] else (idx0[p_, :0], idx1[p_, :0]) for p_ in range(P)]

    def _strided_layout(self, feats1, feats2):
        """(first pointer 1, pair stride 1, first pointer 2, pair stride 2, fp16 pointers or None x 2, keep-alive) if every tensor of the two lists is a block of whole
        64-float rows inside ONE fp32 storage on this device, the blocks of each list equally spaced -- what detectAndCompute's List[Dict] holds; None otherwise.
        (Storage identity, not `_base`: tensors made under torch.inference_mode do not record their view base.)"""
        f0 = feats1[0]
        if not (torch.is_tensor(f0) and f0.is_cuda and f0.dtype == torch.float32 and (self.dev.index is None or f0.device.index == self.dev.index)
                and f0.device.index == torch.cuda.current_device()):      # (self.dev is torch.device('cuda'): no index, never equal to a tensor's cuda:0)
            return None
        st = f0.untyped_storage().data_ptr()
        offs = []
        for lst in (feats1, feats2):
            o = []
            for f in lst:
                if not (torch.is_tensor(f) and f.is_cuda and f.dtype == torch.float32 and f.dim() == 2 and f.shape[1] == 64 and f.untyped_storage().data_ptr() == st
                        and (f.shape[0] <= 1 or f.stride(0) == 64) and f.stride(1) == 1 and f.storage_offset() % 8 == 0):
                    return None
                o.append(f.storage_offset())
            stride = o[1] - o[0] if len(o) > 1 else 64 * max(int(lst[0].shape[0]), 1)
            if stride <= 0 or stride % 8 or any(o[i + 1] - o[i] != stride for i in range(len(o) - 1)) or any(int(f.shape[0]) * 64 > stride for f in lst):
                return None
            offs.append((o[0], stride))
        # No fp16 copies: the matcher's own conversion pass reads the rows as they are NOW (round 5 reused the copies detectAndCompute's descriptor kernel had written,
        # keyed on the storage pointer: a caller who re-scaled or masked the returned descriptors in place got a filter on stale rows -- ADVICE r5; the pass is ~1 % of a step)
        return st + 4 * offs[0][0], offs[0][1], st + 4 * offs[1][0], offs[1][1], None, None, [f0]

    def create_xy(self, h, w, dev):
        y, x = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing='ij')
        xy = torch.cat([x[..., None], y[..., None]], -1).reshape(-1, 2)
        return xy

    def extractDense(self, x, top_k=8_000, _scale_div=1.0):
        if top_k < 1:
            top_k = 100_000_000
        x, rh1, rw1 = self.preprocess_tensor(x)
        lib = _lib.load()
        B, _, H, W = x.shape
        hc, wc = H // 8, W // 8
        feats, _, heat, rel = self.net.backbone(x, want_logits=True, want_heat=False)
        k = min(hc * wc, int(top_k))
        mkpts = torch.empty((B, k, 2), dtype=torch.float32, device=x.device)
        desc = torch.empty((B, k, 64), dtype=torch.float32, device=x.device)
        ws, n = self.net.workspace("dense", lib.xfh_dense_workspace_bytes(B, hc, wc, k))
        _lib.check(lib.xfh_extract_dense(self.net.handle(), _ptr(rel), _ptr(feats), B, hc, wc, k, float(rw1), float(rh1),
                                         float(_scale_div), _ptr(mkpts), _ptr(desc), None, _ptr(ws), n, _stream()),
                   "xfh_extract_dense")
        return mkpts, desc

    def extract_dualscale(self, x, top_k, s1=0.6, s2=1.3):
        self._require_gpu()
        x = x.to(self.dev).float().contiguous()
        B, _, H, W = x.shape
        outs = []
        for s, frac in ((s1, 0.20), (s2, 0.80)):
            Ho, Wo = int(math.floor(H * s)), int(math.floor(W * s))
            xs = _LazyResize(x, Ho, Wo, np.float32(1.0 / s), np.float32(1.0 / s))       # both resizes run inside xfh_backbone_resized
            mk, ft = self.extractDense(xs, int(top_k * frac), _scale_div=s)
            sc = torch.ones(mk.shape[:2], device=mk.device) * (1 / s)
            outs.append((mk, sc, ft))
        mkpts = torch.cat([outs[0][0], outs[1][0]], dim=1)
        sc = torch.cat([outs[0][1], outs[1][1]], dim=1)
        feats = torch.cat([outs[0][2], outs[1][2]], dim=1)
        return mkpts, sc, feats

    def parse_input(self, x):
        if len(x.shape) == 3:
            x = x[None, ...]

        if isinstance(x, np.ndarray):
            x = torch.tensor(x).permute(0, 3, 1, 2) / 255

        return x

    def _parse_input_fast(self, x):
        """parse_input for the matchers: a uint8 numpy image keeps its bytes (the /255 moves into the ingest kernel)."""
        if isinstance(x, np.ndarray) and x.dtype == np.uint8 and len(x.shape) in (3, 4):
            if len(x.shape) == 3:
                x = x[None, ...]
            return _U8Image(torch.from_numpy(np.ascontiguousarray(x)).permute(0, 3, 1, 2), 255.0)
        return self.parse_input(x)
