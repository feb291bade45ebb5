"""hipGraph capture of the sparse hot path for small batches (the realtime_demo.py regime).

At B = 1..2 a VGA frame is ~25 kernel launches of 5-50 us each: the step is launch-bound, not GPU-bound.
`CapturedSparsePipeline` records `detectAndCompute` (+ the MNN match of consecutive frames) once into a HIP
graph (torch.cuda.CUDAGraph on ROCm = hipGraph) on fixed-shape static buffers and replays it per call:
one graph launch instead of ~25 kernel launches, the same kernels, bit-identical results.

    pipe = CapturedSparsePipeline(xfeat, batch=2, height=480, width=640, top_k=4096, match=True)
    out = pipe(frames)          # frames: (2,3,480,640) float32 / uint8 CUDA tensor
    out['keypoints'][b][:out['n_valid'][b]], out['matches'][p] ...

Plateau images that overflow the captured NMS capacity are detected after the replay (the candidate counts are
part of the read-back) and re-run eagerly through `xfeat.detectAndCompute`, so results stay exact.

Buffer lifetime: a captured graph bakes in raw device pointers (workspaces, the packed weight blob).  The pipeline
therefore owns a PRIVATE replica of the model (`XFeat(weights=xfeat.net.state_dict())`: own C handle, own workspaces)
that nothing else calls, so no eager call on the user's `xfeat` -- a larger image, the plateau re-run, a
`load_state_dict` -- can reallocate or free memory the graph still references; the eager fallback runs on the
user's object.  Weights are snapshotted at construction: build a new pipeline after changing them.  As a second
line of defence every replay checks the replica's allocation epoch and re-captures if it ever moved.
"""
import torch


class CapturedSparsePipeline:
    def __init__(self, xfeat, batch, height, width, channels=3, top_k=None, detection_threshold=None, match=True,
                 min_cossim=-1, dtype=torch.float32):
        if height % 32 or width % 32:
            raise RuntimeError('CapturedSparsePipeline needs H and W to be multiples of 32 (no resize inside the graph)')
        if match and batch % 2:
            raise RuntimeError('matching consecutive frames needs an even batch')
        xfeat._require_gpu()
        self.user_xf = xfeat                                    # eager fallback only
        self.xf = type(xfeat)(weights=xfeat.net.state_dict(), top_k=xfeat.top_k, detection_threshold=xfeat.detection_threshold)
        self.B, self.match = batch, match
        self.top_k = xfeat.top_k if top_k is None else top_k
        self.thr = xfeat.detection_threshold if detection_threshold is None else detection_threshold
        self.min_cossim = min_cossim
        dev = xfeat.dev
        self.x = torch.zeros((batch, channels, height, width), dtype=dtype, device=dev)
        self.hw = height * width
        self._capture()

    def _capture(self):
        dev = self.xf.dev
        # warm-up on a side stream (allocates workspaces, sets kernel attributes), then capture
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(2):
                self._body()
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._body()
        self._epoch = self.xf.net._ws_epoch

    def _body(self):
        kp, sc, de, nv, nc, cap, hw, d16 = self.xf._detect_device(self.x, self.top_k, self.thr, want_f16=self.match)[:8] if self.match else \
            self.xf._detect_device(self.x, self.top_k, self.thr) + (None,)
        self.kpts, self.scores, self.desc, self.n_valid, self.n_cand, self.cap = kp, sc, de, nv, nc, cap
        if self.match:
            self.idx0, self.idx1, self.n_match = self.xf.match_pairs_device(de, nv, self.min_cossim, d16)
            self.counts = torch.cat([nv, nc, self.n_match, self.xf.net._status_target[:1]])
        else:
            self.counts = torch.cat([nv, nc, self.xf.net._status_target[:1]])      # (last: the replica's status word of the fp16-pair arithmetic)

    @torch.inference_mode()
    def __call__(self, frames):
        """frames: (B,C,H,W) tensor of the captured shape/dtype.  Returns a dict of device tensors (views of the
        static buffers: valid until the next call) plus the host-side counts."""
        if self.xf.net._ws_epoch != self._epoch:                # somebody used the private replica directly: pointers are stale
            self._capture()
        self.x.copy_(frames, non_blocking=True)
        self.graph.replay()
        c = self.counts.cpu()                                   # the one read-back
        if int(c[-1]):                                          # an activation left the range of the fp16-pair arithmetic (never seen on images): the captured kernels' results
            self.xf.net._status_target[:1].zero_()              # are not valid.  The replica falls back to the fp32-range kernels and is re-captured (the graph bakes the kernel choice
            self.xf.net.fx_range_exceeded(status=1)             # in); this call is answered eagerly by the user's model, whose own check makes the same switch
            self._capture()
            return self._eager(frames)
        c = c[:-1]
        B = self.B
        n_valid, n_cand = c[:B].tolist(), c[B:2 * B]
        if self.cap < self.hw and int(n_cand.max()) > self.cap:
            return self._eager(frames)                           # plateau image: exact re-run with room
        out = {'keypoints': self.kpts, 'scores': self.scores, 'descriptors': self.desc, 'n_valid': n_valid}
        if self.match:
            nm = c[2 * B:].tolist()
            out['n_matches'] = nm
            out['matches'] = [(self.idx0[p, :nm[p]], self.idx1[p, :nm[p]]) for p in range(B // 2)]
        return out

    def _eager(self, frames):
        res = self.user_xf.detectAndCompute(frames, top_k=self.top_k, detection_threshold=self.thr)
        K = self.top_k
        dev = self.xf.dev
        kp = torch.zeros((self.B, K, 2), device=dev); sc = torch.zeros((self.B, K), device=dev); de = torch.zeros((self.B, K, 64), device=dev)
        nv = []
        for b, r in enumerate(res):
            n = r['keypoints'].shape[0]; nv.append(n)
            kp[b, :n], sc[b, :n], de[b, :n] = r['keypoints'], r['scores'], r['descriptors']
        out = {'keypoints': kp, 'scores': sc, 'descriptors': de, 'n_valid': nv}
        if self.match:
            ms = [self.user_xf.match(res[2 * p]['descriptors'], res[2 * p + 1]['descriptors'], min_cossim=self.min_cossim) for p in range(self.B // 2)]
            out['matches'] = ms
            out['n_matches'] = [len(m[0]) for m in ms]
        return out
