"""Throughput runner for a STREAM of frame batches (SURVEY 8 hot path: detectAndCompute + match of consecutive frames, modules/xfeat.py:47-96,150-165).

`FrameStream` keeps `lanes` batches in flight: every lane is an XFeat instance of its own (handle, workspaces), batches go to the lanes round-robin, a
batch's ragged counts travel to pinned host memory by an asynchronous copy and are waited for when the batch is retired (in submission order).  Results are,
bit for bit, the ones XFeat._detect_device / match_pairs_device return for the same batch on a handle with the same options.

Two modes:
  * default (concurrent=False): all lanes queue on ONE HIP stream.  The kernels run exactly as in the synchronous path, one after the other; what is gained
    is the host round trip of the read-back (the GPU never waits for Python).
  * concurrent=True: a HIP stream per lane -- the hardware schedules the convolutions of one batch into the latency-bound tail of the other (NMS
    compaction, top-k, the matcher's refine scan and finalize: ~0.12 ms of a 1.7 ms step).
  Either way the lanes run the library's default kernel mix.  (Round 3 found the split-bf16 key-point head delivering a wrong 16-cell block once in ~10^4
  steps with two streams on the chip; round 4 traced it to instruction-cache misses in a workgroup's first tile -- another stream's kernels evict the code --
  and made the f32-MFMA heads the default of every handle: DESIGN 9.0.  Nothing here depends on it any more.)

  `x` must stay untouched until its ticket is retired: the lane's kernels read it asynchronously (and result() reads it again if a plateau image overflowed the
  NMS candidate list).  A caller that refills one staging buffer per frame needs as many buffers as lanes.

    fs = FrameStream(weights, top_k=4096, lanes=2)
    t0 = fs.submit(batch0); t1 = fs.submit(batch1)        # returns as soon as the work is queued
    r0 = fs.result(t0)                                    # waits for batch0 only; batch1 keeps the GPU busy
    t2 = fs.submit(batch2) ...
"""
import torch

from .xfeat import XFeat


class _Lane:
    def __init__(self, xf, stream):
        self.xf = xf
        self.stream = stream
        self.event = torch.cuda.Event()
        self.ticket = None          # ticket of the batch in flight / not yet retired
        self.dev = None             # (4, B) int32: n_valid, n_candidates, n_matches (first B/2), [3, 0] = the handle's status word for this batch
        self.host = None            # its pinned host mirror
        self.out = None


class FrameStream:
    def __init__(self, weights=None, top_k=4096, detection_threshold=0.05, lanes=2, min_cossim=-1, xfeats=None, concurrent=False):
        """`xfeats`: ready XFeat instances to use as lanes (one per lane, each with its own handle) instead of building them from `weights`.
        concurrent: a HIP stream per lane; default: one stream for all lanes."""
        if xfeats is None:
            kw = {} if weights is None else {"weights": weights}
            xfeats = [XFeat(top_k=top_k, detection_threshold=detection_threshold, **kw) for _ in range(int(lanes))]
        if len(xfeats) < 1:
            raise ValueError("FrameStream needs at least one lane")
        if len({id(x) for x in xfeats}) != len(xfeats):
            raise ValueError("FrameStream: every lane needs an XFeat instance of its own (the workspaces belong to the handle)")
        self.top_k, self.thr, self.min_cossim = int(top_k), float(detection_threshold), min_cossim
        xfeats[0]._require_gpu()          # no GPU / no library: XFeatHipError, never a fallback
        self.concurrent = bool(concurrent) and len(xfeats) > 1
        shared = None if self.concurrent else torch.cuda.Stream()
        self._lanes = [_Lane(x, torch.cuda.Stream() if self.concurrent else shared) for x in xfeats]
        self._next_ticket = 0
        self._next_retire = 0

    @property
    def lanes(self):
        return len(self._lanes)

    @property
    def in_flight(self):
        return self._next_ticket - self._next_retire

    @torch.inference_mode()
    def submit(self, x):
        """Queue detectAndCompute (top_k, detection_threshold) + the MNN match of the frame pairs (2i, 2i+1) of batch x (B even) on the next lane.
        Returns a ticket at once; at most `lanes` tickets may be outstanding (retire with result()).  x is read asynchronously: do not modify it before
        the ticket is retired."""
        ln = self._lanes[self._next_ticket % len(self._lanes)]
        if ln.ticket is not None:
            raise RuntimeError(f"FrameStream: all {len(self._lanes)} lanes are busy; retire ticket {ln.ticket} first (result())")
        B = x.shape[0]
        if B % 2:
            raise ValueError("FrameStream.submit: consecutive frames are matched pairwise, the batch size must be even")
        ln.stream.wait_stream(torch.cuda.current_stream())          # x may have been produced on the caller's stream
        with torch.cuda.stream(ln.stream):
            if ln.dev is None or ln.dev.shape[1] != B:
                ln.dev = torch.zeros((4, B), dtype=torch.int32, device=x.device)
                ln.host = torch.empty((4, B), dtype=torch.int32).pin_memory()
            ln.xf.net.set_status_target(ln.dev[3])                  # the backbone's status bits travel with the counts (no extra read-back; re-registered per
            ln.dev[3, :1].zero_()                                   # call: a handle re-created in between -- load_state_dict -- starts on the model's own word)
            kp, sc, de, nv, nc, cap, hw, d16 = ln.xf._detect_device(x, self.top_k, self.thr, want_f16=True, counts_out=ln.dev[:2])
            i0, i1, nm = ln.xf.match_pairs_device(de, nv, self.min_cossim, d16, n_out=ln.dev[2, :B // 2])
            ln.host.copy_(ln.dev, non_blocking=True)                # the one read-back (ragged results), asynchronous
            ln.event.record(ln.stream)
        ln.out = (kp, sc, de, i0, i1, cap, B, hw, x)
        ln.ticket = self._next_ticket
        self._next_ticket += 1
        return ln.ticket

    def result(self, ticket=None):
        """Wait for the oldest outstanding batch (tickets retire in submission order) and return its results: device tensors keypoints (B,top_k,2),
        scores (B,top_k), descriptors (B,top_k,64), idx0 / idx1 (B/2,top_k) int64, and HOST int32 tensors n_valid (B), n_candidates (B),
        n_matches (B/2) (rows [n:] of the padded device tensors are unspecified, as for XFeat._detect_device)."""
        if self.in_flight == 0:
            raise RuntimeError("FrameStream.result: nothing in flight")
        if ticket is not None and ticket != self._next_retire:
            raise RuntimeError(f"FrameStream.result: tickets retire in order; next is {self._next_retire}, asked for {ticket}")
        ln = self._lanes[self._next_retire % len(self._lanes)]
        ln.event.synchronize()
        cur = torch.cuda.current_stream()
        cur.wait_event(ln.event)
        kp, sc, de, i0, i1, cap, B, hw, x = ln.out
        while True:                                                 # (as detectAndCompute: repeated until neither the range flag nor the candidate capacity asks for it)
            ncmax = int(ln.host[1].max())
            redo = ln.xf.net.fx_range_exceeded(status=int(ln.host[3, 0]))      # fp16-pair arithmetic out of range (never on images): the lane's model is on the fp32-range kernels now
            if not redo and (cap >= hw or ncmax <= cap):
                break
            if ncmax > cap:                                         # a plateau image overflowed the NMS candidate list: exact re-run with room
                cap = min(hw, max(ncmax, 2 * cap))
            with torch.cuda.stream(ln.stream), torch.inference_mode():
                ln.dev[3, :1].zero_()
                kp, sc, de, nv, nc, cap, hw, d16 = ln.xf._detect_device(x, self.top_k, self.thr, cap=cap, want_f16=True, counts_out=ln.dev[:2])
                i0, i1, nm = ln.xf.match_pairs_device(de, nv, self.min_cossim, d16, n_out=ln.dev[2, :B // 2])
                ln.host.copy_(ln.dev, non_blocking=True)
                ln.event.record(ln.stream)
            ln.event.synchronize()
            cur.wait_event(ln.event)
        for t in (kp, sc, de, i0, i1):
            t.record_stream(cur)                                    # allocated on the lane's stream, consumed on the caller's
        res = {"ticket": ln.ticket, "keypoints": kp, "scores": sc, "descriptors": de, "idx0": i0, "idx1": i1,
               "n_valid": ln.host[0].clone(), "n_candidates": ln.host[1].clone(), "n_matches": ln.host[2, :B // 2].clone(), "nms_capacity": cap}
        ln.ticket, ln.out = None, None
        self._next_retire += 1
        return res

    def drain(self):
        """Retire everything in flight; returns the results in order."""
        out = []
        while self.in_flight:
            out.append(self.result())
        return out
