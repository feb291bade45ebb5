"""``InterpolateSparse2d`` on MI355X: drop-in for /root/reference/modules/interpolator.py:10-33.

Same constructor (``mode``, ``align_corners``) and ``forward(x, pos, H, W)`` as the reference; the sampling runs in the
HIP kernel behind ``xfh_sample_sparse`` (include/xfeat_hip.h) with the reference's fp32 coordinate arithmetic
(``2*(pos/(S-1)) - 1``, then grid_sample's un-normalisation, zeros padding).  ``XFeat.interpolator`` is an instance of
this class like in the reference (modules/xfeat.py:37); the detectAndCompute hot path does not call it -- its three
sampling sites are fused into the detection kernels.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib


class InterpolateSparse2d(nn.Module):
    """ Efficiently interpolate tensor at given sparse 2D positions. """

    def __init__(self, mode='bicubic', align_corners=False):
        super().__init__()
        if mode not in _lib.SAMPLE_MODES:
            raise ValueError(f"mode must be one of {sorted(_lib.SAMPLE_MODES)}")
        # like the reference (interpolator.py:11-14,32): the flag is stored and IGNORED -- forward() always samples with align_corners=False
        self.mode = mode
        self.align_corners = align_corners

    def normgrid(self, x, H, W):
        """ Normalize coords to [-1,1] (reference interpolator.py:17-19; kept for API compatibility). """
        return 2. * (x / (torch.tensor([W - 1, H - 1], device=x.device, dtype=x.dtype))) - 1.

    def forward(self, x, pos, H, W):
        """
        Input
            x: [B, C, H, W] feature tensor
            pos: [B, N, 2] tensor of positions
            H, W: int, original resolution of input 2d positions -- used in normalization [-1,1]

        Returns
            [B, N, C] sampled channels at 2d positions
        """
        if not torch.cuda.is_available():
            raise _lib.XFeatHipError("InterpolateSparse2d needs an AMD MI355X (gfx950) GPU; no CPU fallback exists")
        dev = x.device if x.is_cuda else torch.device('cuda', torch.cuda.current_device())
        x = x.to(dev).float().contiguous()
        pos = pos.to(dev).float().contiguous()             # int64 positions: int/int true division gives the same fp32 quotient
        if x.dim() != 4 or pos.dim() != 3 or pos.shape[0] != x.shape[0] or pos.shape[2] != 2:
            raise RuntimeError('expected x (B,C,H,W) and pos (B,N,2)')
        B, Cc, Hm, Wm = x.shape
        N = pos.shape[1]
        out = torch.empty((B, N, Cc), dtype=torch.float32, device=dev)
        if N:
            _lib.check(_lib.load().xfh_sample_sparse(C.c_void_p(x.data_ptr()), C.c_void_p(pos.data_ptr()), B, Cc, Hm, Wm, N, int(H), int(W),
                                                     _lib.SAMPLE_MODES[self.mode], C.c_void_p(out.data_ptr()),
                                                     C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "xfh_sample_sparse")
        return out
