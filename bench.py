#!/usr/bin/env python3
"""Benchmark of the XFeat hot path on MI355X: BASELINE.json's metric
"frames/sec detectAndCompute+match (VGA, top_k=4096)" on its config[1]
(VGA 640x480 sparse, top_k=4096, batch 64 per GPU).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...          # without a launcher: re-runs itself as N ranks (refuses if fewer than N GPUs are visible)

One step = one pass of the hot path over one batch: detectAndCompute on B synthetic VGA
frames already resident in HBM, then the MNN match of the B/2 consecutive frame pairs
(2i, 2i+1) (`match_xfeat` semantics, min_cossim=-1), then the ONE host read-back of the
per-image counts that the ragged results need.  The K timed steps go through
accelerated_features_amd.streaming.FrameStream with `--lanes` (default 2) batches in flight: a step
queues one batch on the next lane (handle + HIP stream) and retires the oldest one -- its counts
arrive by an asynchronous copy -- so that one batch's latency-bound tail runs under the other's
convolutions; nothing is in flight when the timed region starts and the K-th step retires everything
still in flight.  `--lanes 1` = every step waits for its own read-back (`single_lane_synchronous_fps`).
Multi-GPU: every rank is a replica with its
own batch (weak scaling, no data-path collective; RCCL only for the barrier / max-time).

Prints ONE JSON line on rank 0 (see the task contract): value = whole-job frames/s.
`roofline` is measured live with HIP events around the dominant kernel (the largest single
entry of the rocprofv3 kernel stats: block1_mx_kernel) inside the timed region; the match, the two 24->24 convolutions and the
whole MFMA convolution family are reported next to it from short untimed passes of the same step;
`cpu_baseline` times the CPU oracle (a port of the reference's CPU path) on the host cores
for a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from accelerated_features_amd import sharding  # noqa: E402  (shard ranges + the barrier / max-over-ranks timing protocol)

PEAK_MFMA_F32_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak
H, W, TOP_K = 480, 640, 4096
# SURVEY.md 8(d), per VGA frame of this workload: layer-granular compulsory bytes, algorithmic FLOPs (extraction 2.622 GFLOP +
# half a 4096x4096x64 match), and the roofline time sum_k max(bytes_k / 8 TB/s, flops_k / 157.3 TF) = 20.1 us + 6.83 us
ALGO_BYTES_PER_FRAME = 78.6e6
ALGO_FLOPS_PER_FRAME = 2.622e9 + 2.147e9 / 2
T_ROOF_US_PER_FRAME = 26.9


def make_frames(B, seed):
    """B VGA frames, consecutive frames (2i, 2i+1) related by a known shift + noise."""
    import fixtures
    nb = max(1, min(8, B // 2))
    base = fixtures.texture_images(nb, H, W, seed=seed)
    rs = np.random.RandomState(seed + 1)
    frames = []
    for i in range(B // 2):
        a = torch.roll(base[i % nb], shifts=(3 * (i // nb), 5 * (i // nb)), dims=(1, 2))
        if (i // nb) % 2:
            a = a.flip(2)
        b = torch.roll(a, shifts=(16, 24), dims=(1, 2)) + torch.from_numpy((0.02 * rs.randn(3, H, W)).astype(np.float32))
        frames += [a, b]
    if B % 2:
        frames.append(base[0])
    return torch.stack(frames)


def gpu_telemetry(device_index=0):
    """Clock / power / temperature of the GPU this rank runs on, from the amdgpu sysfs nodes (no subprocess: cheap enough to call right before and right after the timed
    region) -- what lets a reader of the line tell a slow BOX from a slow kernel (box-to-box spread of one kernel: +-8 %).  Values the node does not offer are
    absent; no GPU / no sysfs: {"source": "unavailable"}."""
    import glob
    out = {}
    try:
        cards = sorted(d for d in glob.glob("/sys/class/drm/card[0-9]*/device") if os.path.exists(os.path.join(d, "pp_dpm_sclk")))
        if not cards:
            return {"source": "unavailable"}
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES") or ""
        idx = device_index
        try:
            if vis:
                idx = int(vis.split(",")[device_index])
        except (ValueError, IndexError):
            pass
        d = cards[min(idx, len(cards) - 1)]
        out["source"] = d

        def cur_level(name):          # "0: 132Mhz\n1: 2400Mhz *" -> the starred level (MHz)
            try:
                for ln in open(os.path.join(d, name)).read().splitlines():
                    if ln.rstrip().endswith("*"):
                        return int("".join(ch for ch in ln.split(":")[1] if ch.isdigit()))
            except (OSError, ValueError, IndexError):
                return None
            return None
        for key, node in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk"), ("fclk_mhz", "pp_dpm_fclk")):
            v = cur_level(node)
            if v is not None:
                out[key] = v
        for hw in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
            def rd(name, scale):
                try:
                    return round(int(open(os.path.join(hw, name)).read().strip()) * scale, 1)
                except (OSError, ValueError):
                    return None
            for key, node, scale in (("power_w", "power1_average", 1e-6), ("power_w", "power1_input", 1e-6), ("power_cap_w", "power1_cap", 1e-6),
                                     ("temp_edge_c", "temp1_input", 1e-3), ("temp_junction_c", "temp2_input", 1e-3), ("temp_hbm_c", "temp3_input", 1e-3),
                                     ("sclk_now_mhz", "freq1_input", 1e-6), ("mclk_now_mhz", "freq2_input", 1e-6)):
                v = rd(node, scale)
                if v is not None and key not in out:
                    out[key] = v
        try:
            out["busy_percent"] = int(open(os.path.join(d, "gpu_busy_percent")).read().strip())
        except (OSError, ValueError):
            pass
    except Exception as e:          # noqa: BLE001 -- telemetry must never cost the line
        out["error"] = f"{type(e).__name__}: {e}"
    return out


class GpuStateSampler:
    """sclk / power sampled from the amdgpu sysfs nodes in a background thread WHILE a region runs (a read right before or after it sees an idle GPU: 94 MHz): start(),
    stop() -> {"samples": n, "sclk_mhz": [min, median, max], "power_w": [min, median, max]}.  ~0.3 ms per sample, one sample per millisecond; no subprocess."""

    def __init__(self, device_index=0, period_s=0.001):
        import glob
        import threading
        self.period, self._stop, self.samples = period_s, threading.Event(), []
        cards = sorted(d for d in glob.glob("/sys/class/drm/card[0-9]*/device") if os.path.exists(os.path.join(d, "pp_dpm_sclk")))
        self.freq = self.power = None
        if cards:
            d = cards[min(device_index, len(cards) - 1)]
            for hw in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
                f, p = os.path.join(hw, "freq1_input"), next((os.path.join(hw, n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(hw, n))), None)
                if os.path.exists(f):
                    self.freq, self.power = f, p
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        while not self._stop.is_set():
            try:
                f = int(open(self.freq).read()) * 1e-6
                p = int(open(self.power).read()) * 1e-6 if self.power else None
                self.samples.append((f, p))
            except (OSError, ValueError):
                pass
            time.sleep(self.period)

    def start(self):
        if self.freq:
            self.thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self.freq:
            self.thread.join(1.0)
        if not self.samples:
            return {"samples": 0}
        stat = lambda v: [round(min(v), 1), round(sorted(v)[len(v) // 2], 1), round(max(v), 1)]
        out = {"samples": len(self.samples), "sclk_mhz_min_median_max": stat([s[0] for s in self.samples])}
        pw = [s[1] for s in self.samples if s[1] is not None]
        if pw:
            out["power_w_min_median_max"] = stat(pw)
        return out


def cpu_baseline(seconds):
    """Oracle (port of the reference CPU path) on the host cores: same workload shape,
    bounded sample: batches of 4 VGA frames + their 2 pair matches, repeated ~`seconds`."""
    import fixtures
    from oracle import xfeat_oracle as O
    sd = fixtures.synthetic_state_dict(0)
    x = make_frames(4, seed=77)
    ncpu = os.cpu_count() or 1
    # torch's CPU conv path does not scale to hundreds of threads on these small maps (it gets
    # slower): pick the fastest thread count from a short calibration and report it.
    best = None
    for t in sorted({c for c in (8, 16, 32, 64, ncpu) if c <= ncpu}):
        torch.set_num_threads(t)
        O.detect_and_compute(sd, x[:1], top_k=TOP_K)
        t0 = time.perf_counter()
        O.detect_and_compute(sd, x[:2], top_k=TOP_K)
        dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (t, dt)
        if dt > 3 * best[1]:
            break
    threads = best[0]
    torch.set_num_threads(threads)

    def one_batch(xb):
        t0 = time.perf_counter()
        out = O.detect_and_compute(sd, xb, top_k=TOP_K)
        for p in range(len(xb) // 2):
            O.match_mnn(out[2 * p]["descriptors"], out[2 * p + 1]["descriptors"], -1)
        return time.perf_counter() - t0

    frames, t_used, iters = 0, 0.0, 0
    while t_used < 0.7 * seconds or iters < 1:                  # the B = 4 sample is the reported value
        t_used += one_batch(x)
        frames += 4
        iters += 1
    # the CPU path at B in {1, 4, 64} (B = 1: extraction of one frame; B = 64: ONE pass of the benchmark batch)
    by_batch = {"4": round(frames / t_used, 3)}
    n1, t1 = 0, 0.0
    while t1 < 0.1 * seconds or n1 < 1:
        t1 += one_batch(x[:1]); n1 += 1
    by_batch["1"] = round(n1 / t1, 3)
    if seconds >= 10:
        by_batch["64"] = round(64 / one_batch(make_frames(64, seed=77)), 3)
    return {"value": round(frames / t_used, 3), "unit": "frames/s", "cores": threads, "kind": "port", "frames_per_s_by_batch": by_batch,
            "sample": f"CPU ORACLE (oracle/xfeat_oracle.py, a port of the reference's CPU path with hand-written samplers; /root/reference "
                      f"itself is not on the GPU box; on 8 cores of the build container the port runs at 0.91x the unmodified reference, tools/cpu_reference_vs_oracle.py): {iters} x (detect_and_compute on 4 VGA frames, top_k={TOP_K} + 2 MNN matches) "
                      f"= {frames} frames in {t_used:.1f} s, torch CPU threads={threads}; by_batch: B=1 extraction only, B=64 one pass"}


def cpu_sample(seconds, fn, unit, units_per_call, what):
    """Bounded CPU-oracle sample for the secondary workloads: repeat fn() for about `seconds`."""
    ncpu = os.cpu_count() or 1
    threads = min(ncpu, 32)
    torch.set_num_threads(threads)
    n, used = 0, 0.0
    while used < seconds or n < 1:
        t0 = time.perf_counter()
        fn()
        used += time.perf_counter() - t0
        n += 1
    return {"value": round(n * units_per_call / used, 4), "unit": unit, "cores": threads, "kind": "port",
            "sample": f"{n} x ({what}) in {used:.1f} s, torch CPU threads={threads}"}


def conv_family_roofline(xf, step_fn, n=2):
    """Secondary, untimed pass shared by the dense / megadepth workloads: HIP events (xfh_profile_select) around every MFMA
    convolution launch of `n` steps -> achieved TFLOP/s with the direct form's algorithmic FLOPs."""
    from accelerated_features_amd import _lib
    lib, handle = _lib.load(), xf.net.handle()
    lib.xfh_profile_select(handle, _lib.PROF_CONV_MFMA)
    for _ in range(n):
        step_fn()
    torch.cuda.synchronize()
    cn, cms, cfl, cby = C.c_int(), C.c_double(), C.c_double(), C.c_double()
    lib.xfh_profile_read(handle, C.byref(cn), C.byref(cms), C.byref(cfl), C.byref(cby))
    lib.xfh_profile_select(handle, _lib.PROF_NONE)
    ach = (cfl.value / 1e12) / (cms.value / 1e3) if cms.value > 0 else 0.0
    # priced two ways: the algorithmic fp32 FLOPs of the direct form against the fp32 peak (the figure of rounds 1-4; above 1 since the layers run on the fp16 matrix cores),
    # and the FLOPs the fp16-pair kernels EXECUTE (three fp16 MFMAs per product) against the fp16 MFMA peak -- `frac`: the one that speaks about the kernels
    return {"bound": "mfma", "kernel": "every MFMA convolution launch of the step behind block1: conv_bxd_kernel<24,24> / conv_bxs2_kernel<24>, conv_rs64_kernel (64 -> 64 and 128 -> 128 3x3, "
                                       "with and without the fused 1x1; column strips on maps wider than its rings), conv_bx64s2x_kernel, the 128 -> 64 1x1 -- fp32 results on "
                                       "v_mfma_f32_32x32x16_f16 with an fp16 pair per operand (three MFMAs per product)",
            "achieved": round(3 * ach, 2), "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s", "frac": round(3 * ach / PEAK_F16_TFLOPS, 4),
            "achieved_note": "executed fp16 MFMA rate = 3 x the algorithmic fp32 rate (channel / unit padding not counted)",
            "algorithmic_fp32_tflops": round(ach, 2), "algorithmic_vs_fp32_peak": round(ach / PEAK_MFMA_F32_TFLOPS, 4),
            "launches": cn.value, "ms_per_step": round(cms.value / n, 3), "traffic": None}


def load_pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary (PMC passes cannot run inside
    this process: the figure is a constant of the committed profile, stamped with its source)."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(p):
        try:
            d = json.load(open(p))
            return d.get("dominant_kernel_hbm_bytes_per_launch"), f"profiles/pmc_traffic.json ({d.get('collected', 'rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes')}); not re-measured in this run"
        except Exception:
            return None, None
    return None, None


PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK_F16_TFLOPS = 2500.0          # dense fp16 MFMA
MATCH_MAXIMA_BYTES_PER_ENTRY = 2.0      # the matcher's block maxima (k_match_f16.hip): fp16 since round 5


def kernel_roofline_table(spans_us, B, P, n_kpts=TOP_K, fx=1 | 2 | 8 | 2048, block1=7):
    """roofline_kernels: one row per kernel (or tight kernel group) of the step, from the HIP-event spans of an untimed pass
    (xfh_profile_select(XFH_PROF_ALL)).  Per row: us per step; algorithmic HBM bytes (inputs read once + outputs written once) and FLOPs;
    the FLOPs the shipped kernel EXECUTES on the instruction it uses (the fp16-pair arithmetic: 3 MFMAs per product, and the channel / tile padding);
    the peak of that instruction; frac = max(bytes / 8 TB/s, executed / peak) / measured time.  fx / block1: the handle's options (a cleared fx bit = that family on f32 MFMAs)."""
    from accelerated_features_amd.spec import CONVS
    HW = H * W
    px = {"1": B * HW, "2": B * HW // 4, "4": B * HW // 16, "8": B * HW // 64, "16": B * HW // 256, "32": B * HW // 1024}

    def conv_flops(name, out_px):
        c = next(c for c in CONVS if c.name == name)
        return 2.0 * c.cin * c.cout * c.k * c.k * out_px

    f32 = PEAK_MFMA_F32_TFLOPS
    rows = []

    def add(span, kernel, by, fl_alg, fl_exec, peak, pipe, note="", design_bytes=0.0):
        us = spans_us.get(span)
        if us is None or us <= 0:
            return
        t_mem = by / (PEAK_HBM_GBS * 1e3)                    # us
        t_cmp = (fl_exec / 1e12) / peak * 1e6 if peak else 0.0
        rows.append({"kernel": kernel, "us": round(us, 1), "bytes": int(by), "hbm_gbs": round(by / us / 1e3, 1), "flops_algorithmic": fl_alg,
                     "flops_executed": fl_exec, "pipe": pipe, "peak_used_tflops": peak, "bound": "hbm" if t_mem >= t_cmp else pipe,
                     "floor_us": round(max(t_mem, t_cmp), 1), "frac": round(max(t_mem, t_cmp) / us, 3), **({"note": note} if note else {}),
                     # traffic the kernel's DESIGN adds on top of the algorithmic bytes (the matcher's block-maxima arrays): in no floor, listed so that it can be seen
                     **({"filter_design_traffic_bytes": int(design_bytes)} if design_bytes else {})})

    ci = {c.name: i for i, c in enumerate(CONVS)}
    PAIR, F32P = "fp16 mfma x3 (fp16 pair, fp32-equivalent)", "f32 mfma"

    def conv_row(name, kernel_fx, by, fl, exec_factor, on_fx):      # a convolution layer (or fused pair) on its fp16-pair kernel, or -- fx bit cleared -- on conv_mfma_kernel
        if on_fx:
            add(100 + ci[name], kernel_fx, by, fl, fl * exec_factor, PEAK_F16_TFLOPS, PAIR)
        else:
            add(100 + ci[name], f"conv_mfma_kernel ({name}; fp32-range fallback)", by, fl, fl, f32, F32P)

    add(200, "gray_stats + gray_coef (channel mean, InstanceNorm statistics)", 4.0 * px["1"] * 4, 4.0 * px["1"], 4.0 * px["1"], 0, "valu")
    if block1 == 7:      # block1.2, .3 on v_mfma_f32_16x16x32_f16 in the fp16-pair arithmetic: the row keeps the algorithmic FLOPs against the fp32 peak -- the kernel stays bound by the vector stages
        add(3, "block1_mx_kernel (block1.0-.3 + skip1; block1.2, .3 on fp16-pair MFMAs)", 4.0 * (px["1"] + 24 * px["4"]), 720.0 * px["1"], 720.0 * px["1"], f32, "fp32 valu (v_pk_fma_f32) + fp16 mfma x3")
    else:
        add(3, "block1_fused_kernel (block1.0-.3 + skip1; fp32-range fallback)", 4.0 * (px["1"] + 24 * px["4"]), 720.0 * px["1"], 720.0 * px["1"], f32, "fp32 valu (v_pk_fma_f32)")
    c24, c64 = bool(fx & 2), bool(fx & 1)
    for n_ in ("block2.0", "block2.1"):      # executed: 3 MFMAs per product, couts padded 24 -> 32, K groups 27 -> 28
        conv_row(n_, f"conv_bxd_kernel<24,24> ({n_})", 4.0 * 48 * px["4"], conv_flops(n_, px["4"]), 3 * (32 / 24) * (224 / 216), c24)
    conv_row("block3.0", "conv_bxs2_kernel<24> (block3.0, stride 2)", 4.0 * (24 * px["4"] + 64 * px["8"]), conv_flops("block3.0", px["8"]), 3 * (224 / 216), c24)
    for n3, n1, tag in (("block3.1", "block3.2", "conv_rs64_kernel<1>"), ("block_fusion.1", "block_fusion.2", "conv_rs64_kernel<2> (channels-last out)")):
        conv_row(n3, f"{tag} ({n3} + {n1})", 4.0 * 128 * px["8"], conv_flops(n3, px["8"]) + conv_flops(n1, px["8"]), 3, c64)
    conv_row("block_fusion.0", "conv_rs64_kernel<0> (block_fusion.0)", 4.0 * 128 * px["8"], conv_flops("block_fusion.0", px["8"]), 3, c64)
    for n_, cin, cout, sc_in, sc_out in (("block4.0", 64, 64, "8", "16"), ("block5.0", 64, 128, "16", "32")):
        ho, wo = H // int(sc_out), W // int(sc_out)
        pad = (-(-ho // 8) * 8) * (-(-wo // 16) * 16) / float(ho * wo)          # 8x16-pixel units over the map (1.28 / 1.71 at VGA)
        conv_row(n_, f"conv_bx64s2x_kernel<{cout // 64}> ({n_}, stride 2)", 4.0 * (cin * px[sc_in] + cout * px[sc_out]), conv_flops(n_, px[sc_out]), 3 * pad, c64)
    for n_, ch, sc in (("block4.1", 64, "16"), ("block4.2", 64, "16"), ("block5.1", 128, "32"), ("block5.2", 128, "32")):
        conv_row(n_, f"conv_rs64_kernel<0{', 128' if ch == 128 else ''}> ({n_})", 4.0 * 2 * ch * px[sc], conv_flops(n_, px[sc]), 3, c64)
    fl1 = conv_flops("block5.3", px["32"])
    add(100 + ci["block5.3"], "conv_mfma_kernel<128,64,1x1> (block5.3)", 4.0 * (128 + 64) * px["32"], fl1, fl1, f32, F32P)
    if spans_us.get(100 + ci["block5.3"]):      # block5.3 ran as a launch of its own (fp32-range fallback path): the pyramid sum reads its x5
        add(201, "pyramid_sum_kernel (x3 + up(x4) + up(x5))", 4.0 * 64 * (2 * px["8"] + px["16"] + px["32"]), 3.0 * 64 * px["8"], 3.0 * 64 * px["8"], 0, "valu")
    else:                                       # the default: block5.3 inside the pyramid sum (reads block5.2's 128 channels instead of x5; + the 1x1's FLOPs on the vector ALUs)
        add(201, "pyramid53_kernel (block5.3's 1x1 fused in: x3 + up(x4) + up(relu(W y5 + b)))", 4.0 * (64 * (2 * px["8"] + px["16"]) + 128 * px["32"]),
            3.0 * 64 * px["8"] + fl1, 3.0 * 64 * px["8"] + fl1, f32, "fp32 valu (v_pk_fma_f32)")
    fl_rel = 2.0 * (2 * 64 * 64 + 64) * px["8"]
    fl_kp = 2.0 * (3 * 64 * 64 + 64 * 65) * px["8"]
    fl_kp_mx = 2.0 * (3 * 64 * 64 + 64 * 64) * px["8"]      # (the dustbin logit is a vector dot product)
    if fx & 8:
        add(202, "head_bx_kernel<false> (heatmap_head + 1/|feats|)", 4.0 * (64 + 2) * px["8"], fl_rel, fl_rel * 3, PEAK_F16_TFLOPS, PAIR)
        add(203, "head_bx_kernel<true> (keypoint_head + softmax + depth-to-space)", 4.0 * 2 * px["1"], fl_kp, fl_kp_mx * 3, PEAK_F16_TFLOPS, PAIR)
    else:
        add(202, "head_f32r_kernel<false> (heatmap_head + 1/|feats|; fp32-range fallback)", 4.0 * (64 + 2) * px["8"], fl_rel, fl_rel, f32, F32P)
        add(203, "head_f32r_kernel<true> (keypoint_head + softmax + depth-to-space; fp32-range fallback)", 4.0 * 2 * px["1"], fl_kp, fl_kp_mx, f32, F32P)
    add(210, "nms_flags_kernel", 4.0 * px["1"] + px["1"] / 8, 25.0 * px["1"], 25.0 * px["1"], 0, "valu")
    add(211, "nms_compact_kernel", px["1"] / 8 + 4.0 * px["1"] / 64, 0, 0, 0, "latency")
    add(212, "topk_sort_runs + topk_rank_merge", 4.0 * 3 * B * 2 * n_kpts, 0, 0, 0, "lds sort / latency")
    add(213, "descriptor_kernel (bicubic sampling of normalised feats, 16 taps x 256 B per key-point)", 4.0 * 64 * px["8"] + B * n_kpts * (256 + 128 + 12), 2.0 * 16 * 64 * B * n_kpts,
        2.0 * 16 * 64 * B * n_kpts, 0, "l1 gather", note="bytes = compulsory HBM traffic; the gather itself moves 4 KB per key-point through L1/L2")
    mm = 2.0 * P * n_kpts * n_kpts * 64
    add(220, "match: memset of keys / maxima", 8.0 * 2 * P * n_kpts + 4.0 * P * n_kpts, 0, 0, 0, "hbm")
    maxima = MATCH_MAXIMA_BYTES_PER_ENTRY * 2 * P * n_kpts * (n_kpts / 32)      # R (P, N2/32, N1) + C (P, N1/32, N2): written by the sweep, read by the refine's scan
    add(222, "mnn_f16_sweep2_kernel (every tile multiplied once; column-direction block maxima elementwise over the rows that meet in a lane)", 2.0 * 2 * P * n_kpts * 64, mm, mm, PEAK_F16_TFLOPS, "fp16 mfma (filter)", design_bytes=maxima)
    add(223, "mnn_f16_thr_row + mnn_f16_refine_kernel (scan of the block maxima, exact fp32 blocks on f32 mfma)", 4.0 * 2 * P * n_kpts * 64, 0,
        2.0 * 2 * P * n_kpts * 1.07 * 32 * 64, f32, "latency + f32 mfma", design_bytes=maxima)
    add(224, "mnn_finalize_kernel", 8.0 * 2 * P * n_kpts + 16.0 * P * n_kpts, 0, 0, 0, "latency")
    rows.sort(key=lambda r: -r["us"])
    tot = sum(r["us"] for r in rows)
    floor = sum(r["floor_us"] for r in rows)
    by = sum(r["bytes"] for r in rows)
    return rows, {"kernels_us_per_step": round(tot, 1), "sum_of_floors_us_per_step": round(floor, 1), "frac": round(floor / tot, 3) if tot else None,
                  "hbm_bytes_per_step_algorithmic_by_kernel": int(by), "hbm_frac_of_8tbs": round(by / (tot * 1e-6) / 8e12, 3) if tot else None}


def side_workloads(xf, x, B, seconds_cap=90.0):
    """Short passes of the other BASELINE configs and of the public API, outside the contract's timed region, so that the one JSON line of the
    default run carries them (rank 0, one GPU).  Each entry: a rate or {"error": ...}; nothing here can fail the main line."""
    import fixtures
    out = {}
    t_begin = time.perf_counter()

    def timed(fn, n):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    def guarded(key, fn):
        if time.perf_counter() - t_begin > seconds_cap:
            out[key] = {"error": "skipped: side-pass time budget used up"}
            return
        try:
            out[key] = fn()
        except Exception as e:                      # noqa: BLE001  (a side figure must never take the contract line down)
            out[key] = {"error": f"{type(e).__name__}: {e}"[:200]}

    def public_api():
        # public names returning the reference's types: detectAndCompute -> List[Dict] (one read-back), then match_many -> [(idx0, idx1)] per pair (one read-back)
        def step():
            res = xf.detectAndCompute(x, top_k=TOP_K)
            return xf.match_many([r['descriptors'] for r in res[0::2]], [r['descriptors'] for r in res[1::2]], min_cossim=-1)
        return round(B / timed(step, 3), 1)

    def public_api_per_pair():
        # the same with the reference's own per-pair call: one host round trip per pair (round 4's public_api_fps)
        def step():
            res = xf.detectAndCompute(x, top_k=TOP_K)
            return [xf.match(res[2 * p]['descriptors'], res[2 * p + 1]['descriptors'], min_cossim=-1) for p in range(B // 2)]
        return round(B / timed(step, 3), 1)

    def latency_b1():
        # the reference demo's per-frame step through the public API (realtime_demo.py:204-209): detectAndCompute on ONE VGA frame, match(descs_ref, descs_cur, 0.82)
        # against the cached reference frame, the matched coordinates brought to the host; wall clock per frame, one frame in flight (top_k 4096: BASELINE's figure; the demo's own default is 3000)
        ref = xf.detectAndCompute(x[:1], top_k=TOP_K)[0]

        def step():
            cur = xf.detectAndCompute(x[1:2], top_k=TOP_K)[0]
            i0, i1 = xf.match(ref['descriptors'], cur['descriptors'], 0.82)
            return ref['keypoints'][i0].cpu(), cur['keypoints'][i1].cpu()
        for _ in range(10):
            step()
        return round(1e3 * timed(step, 100), 3)

    def dense():
        P = 32
        base = fixtures.texture_images(4, 1024, 1024, seed=2000)
        a = torch.cat([torch.roll(base, (7 * i, 5 * i), (2, 3)) for i in range(P // 4)])[:P].cuda()
        g = torch.Generator(device="cuda").manual_seed(7)
        b = (a + 0.005 * torch.randn(a.shape, device="cuda", generator=g)).contiguous()
        return round(P / timed(lambda: xf.match_xfeat_star(a, b, top_k=TOP_K), 4), 1)

    def megadepth():
        from accelerated_features_amd.batching import match_pairs
        sizes, _ = sharding.megadepth_pair_sizes()
        big = (fixtures.texture_images(2, 1600, 1600, seed=31) * 255).round().clamp(0, 255).to(torch.uint8).cuda()
        img = lambda hw, v: big[v, :, :hw[0], :hw[1]].contiguous()
        pairs = [(img(a_, i % 2), img(b_, (i + 1) % 2) if a_ != b_ else torch.roll(img(a_, i % 2), (8 + i % 5, 16), (1, 2))) for i, (a_, b_) in enumerate(sizes)]
        return round(len(pairs) / timed(lambda: match_pairs(xf, pairs, top_k=TOP_K, min_cossim=-1, max_pairs=16), 1), 1)

    def lighterglue():
        from accelerated_features_amd.lighterglue import LighterGlue
        lg = LighterGlue(weights=fixtures.lighterglue_state_dict(0))

        def step():
            kp, sc, de, nv, nc, cap, hw = xf._detect_device(x, TOP_K, 0.05)
            m, s_, n = lg.match_pairs_device(kp, de, nv, (W, H), 0.0)
            return torch.cat([nv, n]).cpu()
        return round(B / timed(step, 2), 1)

    guarded("public_api_fps", public_api)
    guarded("public_api_per_pair_fps", public_api_per_pair)
    guarded("latency_b1_ms", latency_b1)
    guarded("dense_1024_pairs_per_s", dense)
    guarded("megadepth1600_pairs_per_s", megadepth)
    guarded("lighterglue_frames_per_s", lighterglue)
    out["side_workloads"] = ("public_api_fps: detectAndCompute (List[Dict]) + match_many (the list form of match(): the reference's per-pair tuples from one launch sequence and one read-back) on the bench batch; public_api_per_pair_fps: the same with 32 x match(); latency_b1_ms: ONE VGA frame through detectAndCompute + match(0.82) "
                             "against a cached reference frame + the matched points on the host, wall clock per frame (the reference demo's step, realtime_demo.py:204-209); dense_1024: match_xfeat_star on 32 pairs of 1024^2 "
                             "(configs[2]); megadepth1600: the 1500 pairs of the MegaDepth-1500 list at long side 1600 through batching.match_pairs, one pass "
                             "(configs[3], crops of one synthetic texture); lighterglue: detect + attention matcher on the bench batch (configs[4]); "
                             "each a short untimed-by-the-contract pass on this GPU, full versions: --workload dense|megadepth|lighterglue")
    return out


def bench_dense(args, xf, rank, world, dist):
    """BASELINE configs[2]: match_xfeat_star (semi-dense, dual scale, MNN + refinement) on 1024x1024 pairs."""
    import fixtures
    P = 32 if args.batch == 64 else args.batch
    base = fixtures.texture_images(4, 1024, 1024, seed=2000 + rank)
    a = torch.cat([torch.roll(base, (7 * i, 5 * i), (2, 3)) for i in range(P // 4)])[:P].cuda()
    # pair image = a noisy copy (fixtures.star_pair's recipe): the dual-scale path resizes by 0.6 / 1.3 and the backbone strides by 32,
    # so no non-trivial shift keeps the cells of both scales aligned; a noisy copy gives >= 2000 mutual matches per pair, i.e.
    # refine_matches (fine_matcher MLP on MFMA + subpixel softmax + confidence filter) does its full work inside the timed region
    g = torch.Generator(device="cuda").manual_seed(7 + rank)
    b = (a + 0.005 * torch.randn(a.shape, device="cuda", generator=g)).contiguous()

    def step():
        return xf.match_xfeat_star(a, b, top_k=TOP_K)

    secs, res = sharding.timed_steps(step, args.steps, args.warmup, dist, torch.cuda.synchronize, "cuda")
    tmax = torch.tensor([secs], dtype=torch.float64)
    if rank == 0:
        cpu = None
        if world == 1 and args.cpu_seconds > 0:
            from oracle import xfeat_oracle as O
            sd = fixtures.synthetic_state_dict(0)
            ca, cb = a[:1].cpu(), b[:1].cpu()
            cpu = cpu_sample(args.cpu_seconds, lambda: O.match_xfeat_star(sd, ca, cb, top_k=TOP_K), "pairs/s", 1,
                             "oracle match_xfeat_star on one 1024x1024 pair")
        print(json.dumps({
            "cpu_baseline": cpu, "roofline": conv_family_roofline(xf, step),
            "metric": "image pairs/sec match_xfeat_star (1024x1024, top_k=4096)", "value": round(world * P * args.steps / float(tmax.item()), 2),
            "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * float(tmax.item()) / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "match_xfeat_star semi-dense on 1024x1024 pairs, batch=32 pairs per GPU (BASELINE configs[2])",
                       "pairs_per_gpu": P, "top_k": TOP_K, "pair": "image + 0.005*noise (see bench_dense)",
                       "mean_refined_matches": round(float(np.mean([len(r) for r in res])), 1)}}))
    close_group(dist)


def bench_lighterglue(args, xf, rank, world, dist):
    """BASELINE configs[4]: XFeat + LighterGlue on VGA pairs, batch 64 frames = 32 pairs per GPU.
    One step = detectAndCompute of the 64 frames + the attention matcher on the 32 consecutive pairs (fixed-capacity
    key-point lists with device-side counts: one read-back per step)."""
    import fixtures
    from accelerated_features_amd.lighterglue import LighterGlue
    from accelerated_features_amd import _lib as _lib_mod
    lg = LighterGlue(weights=fixtures.lighterglue_state_dict(0))
    B = args.batch
    x = make_frames(B, seed=1000 + rank).cuda()

    def step():
        kp, sc, de, nv, nc, cap, hw = xf._detect_device(x, TOP_K, 0.05)
        m, s, n = lg.match_pairs_device(kp, de, nv, (W, H), 0.0)     # synthetic matcher weights: keep every mutual assignment
        return torch.cat([nv, n]).cpu()

    secs, counts = sharding.timed_steps(step, args.steps, args.warmup, dist, torch.cuda.synchronize, "cuda")
    tmax = torch.tensor([secs], dtype=torch.float64)
    if rank == 0:
        nv, nm = counts[:B].tolist(), counts[B:].tolist()
        kpt = float(np.mean(nv))
        # algorithmic FLOPs of one pair at n key-points per image (no pruning credit): 6 layers x (4 attentions of
        # 4 n^2 d + linears 2 x 2 n (3dd + dd + 4dd + 2dd + 2dd + dd + 4dd + 2dd)) + the similarity matrix 2 n^2 d
        d = 96
        flops_pair = 6 * (4 * 4 * kpt * kpt * d + 2 * 2 * kpt * 19 * d * d) + 2 * kpt * kpt * d
        out = {"metric": "frames/sec detectAndCompute + LighterGlue (VGA, top_k=4096)", "value": round(world * B * args.steps / float(tmax.item()), 2),
               "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * float(tmax.item()) / args.steps, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "XFeat + LighterGlue attention matcher on VGA pairs, batch=64 frames (32 pairs) per GPU (BASELINE configs[4])",
                          "batch_per_gpu": B, "top_k": TOP_K, "weights": "synthetic (tests/fixtures.py; parity of the matcher unpinned vs kornia, see DESIGN.md)",
                          "mean_keypoints": round(kpt, 1), "mean_matches": round(float(np.mean(nm)), 1),
                          "prune_min_kpts": lg.prune_min_kpts, "gflop_per_pair_unpruned": round(flops_pair / 1e9, 1)}}
        # roofline of the dominant kernel (lg_attention_kernel, 60 % of a pair): untimed pass of 3 pairs without width pruning
        # (so that the live key-point counts equal the capacities the FLOPs are counted at), HIP events on the launch stream
        lib = _lib_mod.load()
        lib.xfh_lg_profile(lg.handle(), 1)
        kp, sc, de, nv, nc, cap, hw = xf._detect_device(x[:6], TOP_K, 0.05)
        lg.match_pairs_device(kp, de, nv, (W, H), 0.0, _lib_mod.LG_NO_PRUNING)
        torch.cuda.synchronize()
        n_l, ms, fl = C.c_int(), C.c_double(), C.c_double()
        lib.xfh_lg_profile_read(lg.handle(), C.byref(n_l), C.byref(ms), C.byref(fl))
        lib.xfh_lg_profile(lg.handle(), 0)
        ach = (fl.value / 1e12) / (ms.value / 1e3) if ms.value > 0 else 0.0
        out["roofline"] = {"bound": "mfma", "kernel": "lg_attention_kernel (flash-style softmax(QK^T)V on v_mfma_f32_32x32x2_f32, both images per launch)",
                           "achieved": round(ach, 2), "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_F32_TFLOPS, 4),
                           "launches": n_l.value, "avg_launch_us": round(1e3 * ms.value / max(n_l.value, 1), 2),
                           "algorithmic": "4*Nq*Nk*96 FLOP per image and launch (4096 x 4096 key-points, pruning off for this pass)", "traffic": None}
        if world == 1 and args.cpu_seconds > 0:
            from oracle import lighterglue_oracle as LGO
            sdl = fixtures.lighterglue_state_dict(0)
            kp, sc, de, nv, nc, cap, hw = xf._detect_device(x[:2], TOP_K, 0.05)
            n0, n1 = int(nv[0]), int(nv[1])
            k0, d0, k1, d1 = kp[0, :n0].cpu(), de[0, :n0].cpu(), kp[1, :n1].cpu(), de[1, :n1].cpu()
            size = torch.tensor([float(W), float(H)])
            out["cpu_baseline"] = cpu_sample(args.cpu_seconds, lambda: LGO.lighterglue_forward(sdl, k0, d0, size, k1, d1, size, min_conf=0.0,
                                                                                               prune=True, prune_min_kpts=lg.prune_min_kpts),
                                             "frames/s", 2, f"oracle LighterGlue on one pair of {n0} x {n1} key-points; extraction excluded")
        print(json.dumps(out))
    close_group(dist)


def bench_megadepth(args, xf, rank, world, dist):
    """BASELINE configs[3]: the MegaDepth-1500 pair list at long side 1600, sharded across the GPUs (no collective).
    Image sizes follow the reference's pair list (tests/golden/megadepth1500_sizes.json, generated from
    assets/megadepth_1500.json), scaled x1600/1184 and floored to multiples of 32; pixels are synthetic
    uint8 textures resident in HBM.  One step = the whole job: this rank's contiguous shard of the 1500 pairs through
    accelerated_features_amd.batching.match_pairs (size-grouped batches, match_xfeat's results per pair)."""
    import fixtures
    from accelerated_features_amd.batching import match_pairs
    shard_range = sharding.shard_range
    sizes, n_distinct = sharding.megadepth_pair_sizes()
    lo, hi = shard_range(len(sizes), rank, world)
    bank = {}

    def image(hw, variant):
        if hw not in bank:
            t = fixtures.texture_images(2, hw[0], hw[1], seed=(hw[0] * 7 + hw[1]) % 1000)
            bank[hw] = (t * 255).round().clamp(0, 255).to(torch.uint8).cuda()
        return bank[hw][variant]

    pairs = []
    for i in range(lo, hi):
        a, b = sizes[i]
        pairs.append((image(a, i % 2), image(b, (i + 1) % 2) if a != b else torch.roll(image(a, i % 2), (8 + i % 5, 16), (1, 2))))

    def step():
        return match_pairs(xf, pairs, top_k=TOP_K, min_cossim=-1, max_pairs=16)

    secs, res = sharding.timed_steps(step, args.steps, args.warmup, dist, torch.cuda.synchronize, "cuda")
    tmax = torch.tensor([secs], dtype=torch.float64)
    if rank == 0:
        t = float(tmax.item())
        mpix = sum(a[0] * a[1] + b[0] * b[1] for a, b in sizes) / 1e6
        cpu = None
        if world == 1 and args.cpu_seconds > 0:
            from oracle import xfeat_oracle as O
            sd = fixtures.synthetic_state_dict(0)
            ca, cb = pairs[0][0][None].cpu().float(), pairs[0][1][None].cpu().float()
            cpu = cpu_sample(args.cpu_seconds, lambda: O.match_xfeat(sd, ca, cb, top_k=TOP_K), "pairs/s", 1,
                             f"oracle match_xfeat on one pair of {tuple(ca.shape[2:])} / {tuple(cb.shape[2:])} images")
        print(json.dumps({
            "cpu_baseline": cpu, "roofline": conv_family_roofline(xf, step, 1),
            "metric": "image pairs/sec match_xfeat over the MegaDepth-1500 pair list (long side 1600, top_k=4096)",
            "value": round(len(sizes) * args.steps / t, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * t / args.steps, 2), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "MegaDepth-1500 pair list, sizes scaled to long side 1600 and /32, sharded contiguously across the GPUs "
                                   "(BASELINE configs[3]); one step = all 1500 pairs",
                       "pairs": len(sizes), "pairs_this_rank": hi - lo, "distinct_size_pairs": n_distinct, "megapixels_per_pass": round(mpix, 1),
                       "top_k": TOP_K, "input": "uint8 tensors in HBM", "mean_matches_rank0": round(float(np.mean([len(r[0]) for r in res])), 1),
                       "parallelism": f"pairs sharded x{world}, no collective", "barrier": getattr(args, "barrier_name", None)}}))
    close_group(dist)


def bench_demo(args, xf, rank, world, dist):
    """SURVEY 8(f) f4 -- the per-frame work of the reference demo (realtime_demo.py:204-231) for B streams per GPU: reference features cached,
    each step = detectAndCompute of the B current frames + mutual-NN match against the cached descriptors + MAGSAC++ homography per stream."""
    from accelerated_features_amd.homography import ReferenceTracker
    B = args.batch
    import fixtures
    rs = np.random.RandomState(77 + rank)
    base = fixtures.texture_images(min(B, 8), H, W, seed=1000 + rank)
    ref = torch.stack([torch.roll(base[i % len(base)], shifts=(7 * (i // len(base)), 11 * (i // len(base))), dims=(1, 2)) for i in range(B)])
    # current frame = reference translated by (64, 32) px (a multiple of the backbone's stride: the synthetic weights' descriptors are not
    # trained for anything else) + noise; min_cossim -1 instead of the demo's 0.82 for the same reason: ~1/3 of the list are wrong matches
    cur = (torch.roll(ref, shifts=(32, 64), dims=(2, 3)) + torch.from_numpy((0.01 * rs.randn(B, 3, H, W)).astype(np.float32))).cuda()
    ref = ref.cuda()
    tr = ReferenceTracker(xf, top_k=TOP_K, min_cossim=-1, ransac_thr=4.0, min_inliers=50, max_iters=700, confidence=0.995)
    tr.set_reference(ref)

    def step():
        return tr.track(cur)

    secs, r = sharding.timed_steps(step, args.steps, args.warmup, dist, torch.cuda.synchronize, "cuda")
    tmax = torch.tensor([secs], dtype=torch.float64)
    if rank == 0:
        info = r["info"].cpu().numpy()
        Hm = r["H"].cpu().numpy()
        err = float(np.abs(Hm - np.array([[1, 0, 64.0], [0, 1, 32.0], [0, 0, 1]])).max())
        # the homography stage alone (three launches), HIP events on the launch stream
        from accelerated_features_amd.homography import find_homography_matches
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            find_homography_matches(tr.ref[0], r["keypoints"], r["idx0"], r["idx1"], r["n_matches"], 4.0, 700, 0.995, 0)
        e1.record()
        torch.cuda.synchronize()
        hg_us = e0.elapsed_time(e1) * 100.0
        # one stream (the demo itself): latency of a step at B = 1
        tr1 = ReferenceTracker(xf, top_k=TOP_K, min_cossim=-1)
        tr1.set_reference(ref[:1])
        for _ in range(3):
            tr1.track(cur[:1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            tr1.track(cur[:1])
        torch.cuda.synchronize()
        lat1 = (time.perf_counter() - t0) / 20
        cpu = None
        if world == 1 and args.cpu_seconds > 0:
            from oracle import homography_oracle as HO
            k = int(r["n_matches"][0])
            p0 = tr.ref[0][0][r["idx0"][0, :k]].cpu().numpy()
            p1 = r["keypoints"][0][r["idx1"][0, :k]].cpu().numpy()
            cpu = cpu_sample(min(args.cpu_seconds, 10.0), lambda: HO.find_homography(p0, p1, 4.0, 700, 0.995), "frames/s", 1,
                             f"numpy restatement of the MAGSAC++ homography stage alone on one list of {k} matches; extraction and matching excluded")
        nm = float(r["n_matches"].float().mean())
        print(json.dumps({
            "cpu_baseline": cpu,
            "roofline": {"bound": "latency", "kernel": "homography stage: homog_tables + homog_score (hypotheses 0-255) + homog_bound + homog_score (rest, bounded) + homog_select",
                         "avg_call_us": round(hg_us, 1), "pairs_per_call": B, "achieved": None, "peak": None, "frac": None, "traffic": None,
                         "note": "fp64 VALU scoring (~45 ops per hypothesis x correspondence) + a per-pair latency chain in homog_select; per-kernel times in profiles/r02_f_demo_kernel_stats.csv"},
            "metric": "frames/sec detectAndCompute + match vs cached reference + MAGSAC++ homography (VGA, top_k=4096)",
            "value": round(world * B * args.steps / float(tmax.item()), 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * float(tmax.item()) / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (extraction, matching) + f64 (homography)", "data": "synthetic",
            "config": {"workload": "reference demo step (realtime_demo.py:204-231) for B streams per GPU: cached reference features, detect + match + homography",
                       "streams_per_gpu": B, "top_k": TOP_K, "ransac_thr": 4.0, "max_iters": 700, "confidence": 0.995, "min_cossim": -1,
                       "mean_matches": round(nm, 1), "mean_inliers": round(float(info[:, 3].mean()), 1), "found": int(info[:, 0].sum()),
                       "mean_loop_iterations": round(float(info[:, 2].mean()), 1), "max_abs_H_error_vs_true_shift": round(err, 4),
                       "latency_one_stream_ms": round(1e3 * lat1, 3)}}))
    close_group(dist)


def launch_selftest(args, rank, world):
    """The rank side of `python bench.py --gpus N --launch-selftest` (no GPU touched): the same rendezvous, barrier / exactly-K-steps /
    max-over-ranks protocol and one-line report as the real run, with gloo and a sleep; rank 0 prints n_gpus = the world size it sees."""
    dist, note = (None, None)
    if world > 1:      # the same bring-up as the real run (open_group: probe, agreement through files, fallback), with gloo in RCCL's place
        dist, note = sharding.open_group("gloo", rank, world, None, probe_timeout_s=15.0 if "XFH_TEST_FAIL_COLLECTIVE" in os.environ else 60.0, force_host=args.barrier == "host",
                                         fail_probe=os.environ.get("XFH_TEST_FAIL_COLLECTIVE") == str(rank))
    secs, last = sharding.timed_steps(lambda: time.sleep(0.01 * (rank + 1)) or rank, args.steps, args.warmup, dist, None, "cpu")
    if rank == 0:
        print(json.dumps({"metric": "launch self-test (no GPU work)", "value": round(sharding.aggregate_rate(1, args.steps, world, secs, "weak"), 3), "unit": "steps/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * secs / args.steps, 3), "selftest": True, "scaling": "weak",
                          "barrier": barrier_name(dist, note, "gloo"),
                          "config": {"workload": "sleep of 10 ms x (rank + 1) per step", "units_per_rank_per_step": 1}}))
    close_group(dist)


def barrier_name(dist, note, backend):
    """what synchronised the ranks of this run: the collective backend, or the host-side file barrier and why"""
    if dist is None:
        return "none (one rank)"
    return f"file ({note})" if isinstance(dist, sharding.HostGroup) else backend


def close_group(dist):
    if dist is None:
        return
    sharding.sync_barrier(dist)
    dist.destroy_process_group()
    if getattr(dist, "abandoned_backend", False):      # a collective library that never came up may hang in its own teardown: the line is out, leave
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="frames per GPU per step (BASELINE config: 64)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="bound of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--lanes", type=int, default=2, help="batches in flight per GPU (accelerated_features_amd.streaming.FrameStream: one handle + HIP stream each); "
                                                         "1 = every step waits for its own read-back")
    ap.add_argument("--concurrent-lanes", type=int, default=1, help="1 (default): a HIP stream per lane (FrameStream(concurrent=True)): one batch's convolutions fill the other's "
                                                                        "latency-bound tail; 0: all lanes on one stream (the kernels run one after the other as in the synchronous path)")
    ap.add_argument("--wake-ms", type=float, default=200.0, help="untimed steps for this many ms before the profiling passes and the warm-up (a cold GPU's first ~150 ms run 3-4 %% slow); 0 = none")
    ap.add_argument("--no-side-passes", action="store_true", help="skip the extraction-only / with-H2D side figures (profiling runs)")
    ap.add_argument("--workload", default="sparse", choices=["sparse", "dense", "lighterglue", "megadepth", "demo"],
                    help="sparse = BASELINE configs[1] (the contract metric); dense = configs[2], match_xfeat_star on 1024^2 pairs, batch 32; "
                         "megadepth = configs[3], the MegaDepth-1500 pair list sharded across the GPUs; lighterglue = configs[4]; "
                         "demo = the reference demo's per-frame step (cached reference, match, MAGSAC++ homography; SURVEY 8f f4)")
    ap.add_argument("--barrier", default="rccl", choices=["rccl", "host"],
                    help="what synchronises the ranks of --gpus N > 1 (start / stop barrier, max-over-ranks time; the data path has no collective): rccl (default; if its bring-up "
                         "fails or hangs on any rank, every rank falls back to the host-side file barrier and the line says so) or host (accelerated_features_amd.sharding.HostGroup)")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="CPU self-test of the multi-rank launch path (tests/test_sharding_gloo.py): gloo instead of RCCL, a sleep instead of the hot path")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started without a launcher: become the launcher.  One process per GPU under torch.distributed.run; fewer visible GPUs than asked
        # for is an error, never a line with a smaller n_gpus.
        visible = None if args.launch_selftest else (torch.cuda.device_count() if torch.cuda.is_available() else 0)
        raise SystemExit(sharding.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus, visible))
    rank, local_rank, world = sharding.rank_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if args.launch_selftest:
        return launch_selftest(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dist, group_note = None, None
    if world > 1 or os.environ.get("XFH_BENCH_FORCE_DIST") == "1":      # (the env switch exercises the RCCL path with one rank)
        dist, group_note = sharding.open_group("nccl", rank, world, torch.device("cuda", local_rank), force_host=args.barrier == "host")      # backend "nccl" IS RCCL on ROCm
        if rank == 0 and group_note:
            print(f"# {group_note}", file=sys.stderr, flush=True)
    args.barrier_name = barrier_name(dist, group_note, "rccl")

    import fixtures
    from accelerated_features_amd import XFeat, _lib
    xf = XFeat(weights=fixtures.synthetic_state_dict(0), top_k=TOP_K, detection_threshold=0.05)
    lib = _lib.load()
    if args.workload == "dense":
        return bench_dense(args, xf, rank, world, dist)
    if args.workload == "lighterglue":
        return bench_lighterglue(args, xf, rank, world, dist)
    if args.workload == "megadepth":
        return bench_megadepth(args, xf, rank, world, dist)
    if args.workload == "demo":
        return bench_demo(args, xf, rank, world, dist)
    B = args.batch
    x_host = make_frames(B, seed=1000 + rank)
    x = x_host.cuda()                                      # inputs resident in HBM before the timed region
    handle = xf.net.handle()

    cnt_dev = torch.zeros((3, B), dtype=torch.int32, device="cuda")      # rows: n_valid, n_candidates, n_matches (first B/2 entries)

    def step():
        # ONE batch, synchronously, on this handle (the profiled / side passes below; with --lanes 1 also the timed step):
        # (the descriptor kernel also emits the fp16 copy the matcher's filter sweep reads: no separate conversion pass; all counts
        # land in one buffer: one read-back, no concatenation kernel)
        r = xf.detectAndComputePadded(x, TOP_K, 0.05, counts_out=cnt_dev[:2])          # (public names only: the padded form of detectAndCompute + the pair matcher)
        i0, i1, nm = xf.match_pairs_device(r['descriptors'], r['n_valid'], -1, r['descriptors_f16'], n_out=cnt_dev[2, :B // 2])
        c = cnt_dev.cpu()                                  # the one read-back (ragged results)
        return torch.cat([c[0], c[1], c[2, :B // 2]]), r['nms_capacity']

    # The timed step: the same batch through accelerated_features_amd.streaming.FrameStream -- `lanes` batches in flight, one handle + HIP stream
    # each; a call queues one batch and retires the oldest one once every lane is busy (its ragged counts arrive by an asynchronous copy).  The
    # latency-bound tail of one batch (NMS compaction, top-k, refine scan, finalize, read-back) fills with the convolutions of the next.
    from accelerated_features_amd.streaming import FrameStream
    lanes = max(1, args.lanes)
    # (every lane runs the library's default kernel mix; lane 0 is `xf`, which also serves the single-lane profiling passes after the timed region)
    conc = bool(args.concurrent_lanes) and lanes > 1
    lane_xf = [xf] + [XFeat(weights=fixtures.synthetic_state_dict(0), top_k=TOP_K, detection_threshold=0.05) for _ in range(lanes - 1)]
    handles = [l.net.handle() for l in lane_xf]
    fs = FrameStream(xfeats=lane_xf, top_k=TOP_K, detection_threshold=0.05, min_cossim=-1, concurrent=conc)
    retired = []

    timed_calls = [None]                                   # calls left in the timed region (None: warm-up)

    def lane_step():
        r = None
        if fs.in_flight == fs.lanes:
            r = fs.result()
            retired.append(r)
        fs.submit(x)
        if timed_calls[0] is not None:
            timed_calls[0] -= 1
            if timed_calls[0] == 0:                        # the last timed step also retires what is still in flight: all K results are delivered inside the region
                retired.extend(fs.drain())
        return r

    def arm(last):
        for r in fs.drain():                               # nothing of the warm-up is in flight when the timed region starts
            retired.append(r)
        assert retired and all(int(r["n_candidates"].max()) <= r["nms_capacity"] for r in retired), "NMS capacity overflow in the benchmark workload"
        retired.clear()
        timed_calls[0] = args.steps
        for h_ in handles:
            lib.xfh_profile_select(h_, _lib.PROF_BLOCK1)

    def read_prof(hs=None):
        tot = [0, 0.0, 0.0, 0.0]
        for h_ in (hs or [handle]):
            n_, ms_, fl_, by_ = C.c_int(), C.c_double(), C.c_double(), C.c_double()
            lib.xfh_profile_read(h_, C.byref(n_), C.byref(ms_), C.byref(fl_), C.byref(by_))
            tot = [tot[0] + n_.value, tot[1] + ms_.value, tot[2] + fl_.value, tot[3] + by_.value]
        return tuple(tot)

    def side_prof(which, n=3):            # secondary (untimed) pass: same events mechanism on another kernel family
        lib.xfh_profile_select(handle, which)
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        r = read_prof()
        lib.xfh_profile_select(handle, _lib.PROF_NONE)
        return r


    def span_pass(n=3):                   # every kernel of the step as its own HIP-event span (untimed pass)
        lib.xfh_profile_select(handle, _lib.PROF_ALL)
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        ids, sms, cnt_ = (C.c_int * 8192)(), (C.c_double * 8192)(), C.c_int()
        lib.xfh_profile_read_spans(handle, ids, sms, 8192, C.byref(cnt_))
        lib.xfh_profile_select(handle, _lib.PROF_NONE)
        acc = {}
        for i in range(min(cnt_.value, 8192)):
            acc[ids[i]] = acc.get(ids[i], 0.0) + sms[i]
        return {k: 1e3 * v / n for k, v in acc.items()}

    # GPU wake-up BEFORE the warm-up.  A GPU that has just been woken delivers 3-4 % less for its first ~150 ms of work (power state: bench.py --warmup 5 alone
    # 38.8 k frames/s, --warmup 40 40.2 k, every later window of the same run 40-41.8 k; a kernel span measured cold read 386 us for block1 against 245):
    # untimed windows of steps (at least --wake-ms, until two windows in a row agree within 1 %) bring it to the state a stream of batches runs in; block1's HIP events are created here (hipEventCreate inside
    # the timed region cost its first window 1.8 %).  Then the contract's W warm-up steps and K timed steps, back to back.
    for h_ in handles:
        lib.xfh_profile_select(h_, _lib.PROF_BLOCK1)
    # sclk / power are sampled WHILE THE WAKE-UP WINDOWS RUN (the same steps as the timed region, untimed), not inside the timed region: a thread that reads the SMU's sysfs nodes
    # once per millisecond beside the thread that launches 88 kernels per millisecond cost the timed steps 2-4 % (value against the windows measured right after it, which had no
    # sampler: 53.3 / 54.6 k, 52.7 / 54.6, 53.6 / 55.7 and once 48.6 / 55.8; round 5, without a sampler: 49.8 / 49.7-50.5)
    tele_before = gpu_telemetry(local_rank)
    sampler = GpuStateSampler(local_rank, period_s=0.005).start() if args.wake_ms > 0 else None
    t_wake, n_wake, wake_rates = time.perf_counter(), 0, []
    while args.wake_ms > 0:                                # windows of `steps` untimed steps until two in a row agree within 1 % (and >= wake_ms, <= 8 x wake_ms have passed)
        torch.cuda.synchronize()
        t0w = time.perf_counter()
        for _ in range(args.steps):
            lane_step()
        fs.drain()
        retired.clear()                                    # (window by window: 160 retained results are 11 GB whose release right in front of the warm-up is an idle gap)
        torch.cuda.synchronize()
        wake_rates.append(B * args.steps / (time.perf_counter() - t0w))
        n_wake += args.steps
        el = (time.perf_counter() - t_wake) * 1e3
        settled = len(wake_rates) >= 2 and abs(wake_rates[-1] / wake_rates[-2] - 1.0) < 0.01
        if (el >= args.wake_ms and settled) or el >= 8 * args.wake_ms:
            break
    retired.clear()
    if sampler:
        sampler._stop.set()                                # (told to end; joined behind the timed region: no wait between the wake-up and the warm-up)
    for h_ in handles:
        lib.xfh_profile_select(h_, _lib.PROF_NONE)

    # (the barrier + torch.cuda.synchronize() that closes the timed region waits for every lane: all `steps` batches complete inside it)
    # (nothing between the wake-up windows and the warm-up: the single reads of the sysfs nodes -- a dozen SMU queries with the GPU idle -- used to sit here and cost the timed
    #  region the state the wake-up had just established: value 3.5 % below the windows measured before and after it.  `before` is now read in front of the wake-up.)
    dt_max, _ = sharding.timed_steps(lane_step, args.steps, args.warmup, dist, torch.cuda.synchronize, "cuda", before_timed=arm)
    tele_after = gpu_telemetry(local_rank)
    tele_during = sampler.stop() if sampler else {"samples": 0}
    timed_calls[0] = None
    assert len(retired) == args.steps and fs.in_flight == 0
    last = retired[-1]
    counts = torch.cat([last["n_valid"], last["n_candidates"], last["n_matches"]])
    cap = last["nms_capacity"]

    n_l, ms, fl, by = read_prof(handles)                  # block1 launches of the timed region, every lane (their durations include the other lane's company)
    for h_ in handles:
        lib.xfh_profile_select(h_, _lib.PROF_NONE)
    # untimed profiling passes (12 single-lane steps): every kernel as a span, then three kernel families
    spans_us = span_pass()
    m_n, m_ms, m_fl, m_by = side_prof(_lib.PROF_MATCH)
    b_n, b_ms, b_fl, b_by = side_prof(_lib.PROF_CONV_24_24)
    cn, cms, cfl, cby = side_prof(_lib.PROF_CONV_MFMA)

    # five more timed windows of `steps` steps each (same protocol, this rank only): the spread of the headline figure
    rep_fps = []
    if rank == 0 and not args.no_side_passes:
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                lane_step()
            torch.cuda.synchronize()
            rep_fps.append(B * args.steps / (time.perf_counter() - t0))
            fs.drain()
        retired.clear()

    # SURVEY 8(d) side figures, each its own short pass OUTSIDE the timed region above (rank-local, per GPU)
    def rate(fn, n=5):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return B * n / (time.perf_counter() - t0)

    def extract_only():
        kp, sc, de, nv, nc, cap_, hw = xf._detect_device(x, TOP_K, 0.05)
        return torch.cat([nv, nc]).cpu()

    side = {}
    if rank == 0 and not args.no_side_passes:
        side["extraction_only_fps"] = round(rate(extract_only), 1)
        side["single_lane_synchronous_fps"] = round(rate(step, 20), 1)      # one handle, one stream, every step waits for its read-back (the contract value of rounds 1-3a)
        if lanes > 1:                                                       # for information: the OTHER lane mode (one stream for all lanes / a stream per lane), 60 steps after 20
            cxf = [XFeat(weights=fixtures.synthetic_state_dict(0), top_k=TOP_K, detection_threshold=0.05) for _ in range(2)]
            cfs = FrameStream(xfeats=cxf, top_k=TOP_K, detection_threshold=0.05, min_cossim=-1, concurrent=not conc)

            def crun(n):
                torch.cuda.synchronize(); t0c = time.perf_counter()
                for _ in range(n):
                    if cfs.in_flight == cfs.lanes: cfs.result()
                    cfs.submit(x)
                cfs.drain(); torch.cuda.synchronize()
                return B * n / (time.perf_counter() - t0c)
            crun(20)
            side["lanes_on_one_stream_fps" if conc else "concurrent_lanes_fps"] = round(crun(60), 1)
            del cfs, cxf
        xh32 = x_host.pin_memory()
        xh8 = (x_host * 255).round().clamp(0, 255).to(torch.uint8).pin_memory()

        def with_h2d(src):
            xd = torch.empty(src.shape, dtype=src.dtype, device="cuda")          # (one device buffer: a fresh 236 MB allocation per step would time the allocator)

            def f():
                xd.copy_(src, non_blocking=True)
                kp, sc, de, nv, nc, cap_, hw, d16 = xf._detect_device(xd, TOP_K, 0.05, want_f16=True)
                i0, i1, nm = xf.match_pairs_device(de, nv, -1, d16)
                return torch.cat([nv, nc, nm]).cpu()
            return f
        side["with_h2d_fp32_fps"] = round(rate(with_h2d(xh32)), 1)
        side["with_h2d_uint8_fps"] = round(rate(with_h2d(xh8)), 1)

        def pipelined(src, n=6):
            """the upload of batch i+1 on a copy stream under the step of batch i (two device buffers): what a feeding loop would do"""
            cs, main = torch.cuda.Stream(), torch.cuda.current_stream()
            dev = [torch.empty(src.shape, dtype=src.dtype, device="cuda") for _ in range(2)]
            ready = [torch.cuda.Event() for _ in range(2)]
            free = [torch.cuda.Event() for _ in range(2)]

            def upload(i):
                with torch.cuda.stream(cs):
                    if i >= 2:
                        cs.wait_event(free[i % 2])
                    dev[i % 2].copy_(src, non_blocking=True)
                    ready[i % 2].record(cs)
            upload(0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                upload(i + 1)
                main.wait_event(ready[i % 2])
                kp, sc, de, nv, nc, cap_, hw, d16 = xf._detect_device(dev[i % 2], TOP_K, 0.05, want_f16=True)
                i0, i1, nm = xf.match_pairs_device(de, nv, -1, d16)
                free[i % 2].record(main)
                torch.cat([nv, nc, nm]).cpu()
            torch.cuda.synchronize()
            return B * n / (time.perf_counter() - t0)
        side["with_h2d_fp32_pipelined_fps"] = round(pipelined(xh32), 1)
        side["with_h2d_uint8_pipelined_fps"] = round(pipelined(xh8), 1)

    if rank == 0 and world == 1 and not args.no_side_passes:
        side.update(side_workloads(xf, x, B))
    if rank == 0:
        n_valid = counts[:B].tolist()
        n_match = counts[2 * B:].tolist()
        opt_fx, opt_b1 = C.c_int(), C.c_int()
        lib.xfh_get_option(handle, b"fx", C.byref(opt_fx)); lib.xfh_get_option(handle, b"block1", C.byref(opt_b1))
        k_rows, k_sum = kernel_roofline_table(spans_us, B, B // 2, fx=opt_fx.value, block1=opt_b1.value)
        achieved = (fl / 1e12) / (ms / 1e3) if ms > 0 else 0.0
        fps = sharding.aggregate_rate(B, args.steps, world, dt_max, "weak")
        fps_gpu = fps / world
        traffic, traffic_src = load_pmc_traffic()
        alone_us = spans_us.get(3) or (1e3 * ms / max(n_l, 1))      # block1's launch with the chip to itself (single-lane side pass); without that pass: the timed region's
        out = {
            "metric": "frames/sec detectAndCompute+match (VGA, top_k=4096)",
            "value": round(fps, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt_max / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "VGA 640x480 sparse top_k=4096, batch=64 per GPU: detectAndCompute + MNN match of "
                                   "the 32 consecutive frame pairs (BASELINE configs[1])",
                       "batch_per_gpu": B, "height": H, "width": W, "top_k": TOP_K, "weights": "synthetic (tests/fixtures.py)",
                       "arithmetic": "fp32 results; the 24-, 64- and 128-channel 3x3 convolutions, block1.2/.3 and both heads on fp16 MFMAs with an fp16 PAIR per operand "
                                     "(x = xh + 2^-11 xl: three MFMAs per product, error <= an fp32 direct convolution's; range-guarded, the f32-MFMA / vector-ALU kernels as fallback); "
                                     "the matcher decides on exact fp32 dot products (fp16 MFMAs only pre-select 32-wide blocks inside a derived error window)",
                       "parallelism": f"replicas x{world}, no collective; {lanes} batches in flight per GPU (FrameStream), "
                                      + ("a HIP stream per lane" if conc else "all lanes on one HIP stream"),
                       "barrier": args.barrier_name,
                       "concurrent_lanes": conc,
                       "lanes": lanes,
                       "untimed_before_the_timed_region": f"GPU wake-up ({n_wake} steps: windows of {args.steps} until two agree within 1 %, >= {args.wake_ms:.0f} ms; a cold GPU runs its first ~150 ms 3-4 % slow), then the W warm-up steps",
                       "wake_up_window_fps": [round(r, 1) for r in wake_rates],
                       # the box's state around the timed region (amdgpu sysfs): a slow box reads differently from a slow kernel
                       "gpu_state": {"during_the_untimed_wake_up_windows": tele_during, "before_the_wake_up": tele_before, "after_timed_region": tele_after,
                                     "note": "before / after are single reads behind a synchronisation (an idle GPU: low sclk); `during` is sampled every 5 ms while the untimed wake-up windows run (the same steps as the timed region; a sampler inside the timed region cost it 2-4 %)"},
                       "mean_keypoints": round(float(np.mean(n_valid)), 1), "mean_matches": round(float(np.mean(n_match)), 1)},
            # block1 x4 + skip1 in one kernel: fp32 FMA work on the vector ALUs (v_pk_fma_f32), LDS-tiled.  Neither HBM nor the matrix
            # cores bound it: its vector stages (conv1 recomputed inside conv2: DESIGN 3.2) do; it is priced against the dense fp32 rate of the chip,
            # which is the same 157.3 TFLOP/s for the packed vector FMA and for the f32 MFMA.  The contract's "bound" takes "hbm" | "mfma": "mfma" here names that dense
            # fp32 peak; "bound_detail" says which pipe it is.
            "roofline": {"bound": "mfma", "bound_detail": "fp32 valu (v_pk_fma_f32) + fp16 mfma x3" if opt_b1.value == 7 else "fp32 valu (v_pk_fma_f32)",
                         "kernel": ("block1_fused_kernel (block1.0-.3 + skip1 fused: 3x3 convs 1->4, 4->8 s2, 8->8, 8->24 s2 on v_pk_fma_f32, "
                                    "LDS-tiled; one launch per step)") if opt_b1.value != 7 else
                                   ("block1_mx_kernel (block1.0-.3 + skip1 fused, LDS-tiled, one launch per step: 1->4 and 4->8 s2 on v_pk_fma_f32, "
                                    "8->8 and 8->24 s2 on v_mfma_f32_16x16x32_f16 in the fp16-pair arithmetic = fp32 results; priced as in every round: "
                                    "algorithmic fp32 FLOPs against the dense fp32 peak)"),
                         # frac / achieved / avg_launch_us: the kernel with the chip to itself (single-lane pass of this run, HIP events on the launch stream = rocprofv3's
                         # per-dispatch duration in profiles/*_kernel_stats_1lane.csv) -- the figure that speaks about the kernel, comparable across rounds.  in_situ: the same
                         # launch inside the timed region, where with two lanes it shares the chip with the other lane's kernels (profiles/*_kernel_stats.csv)
                         "achieved": round(fl / max(n_l, 1) / 1e6 / alone_us, 3), "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(fl / max(n_l, 1) / 1e6 / alone_us / PEAK_MFMA_F32_TFLOPS, 4),
                         "avg_launch_us": round(alone_us, 2), "measured": "single-lane pass of this run" if spans_us.get(3) else "timed region",
                         "in_situ": {"avg_launch_us": round(1e3 * ms / max(n_l, 1), 2), "achieved": round(achieved, 3), "frac": round(achieved / PEAK_MFMA_F32_TFLOPS, 4),
                                     "launches": n_l, "lanes": lanes, "streams": lanes if conc else 1},
                         "flops_per_launch": fl / max(n_l, 1),
                         "algorithmic": "720 FLOP per input pixel x B*H*W pixels per launch",
                         "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_algorithmic": int(by / max(n_l, 1))},
            # xfh_match_mnn = fp16 MFMA filter (derived error window, one sweep, every tile multiplied once) + exact fp32 refine of the flagged
            # 32-wide blocks (~1.07 per row and column) on f32 MFMAs; identical match lists to the exact f32 MFMA kernel (tests; option
            # match_exact selects the latter).  "algorithmic" prices the fp32 work of D1.D2^T against the f32 MFMA peak: above 1 means the
            # filter beat an exact f32 GEMM at its peak.
            "roofline_match": {"bound": "mfma", "kernels": "mnn_f16_sweep2_kernel + mnn_f16_thr_kernel + mnn_f16_refine_kernel (the fp16 copies come from the descriptor kernel)",
                               "us_per_step": round(1e3 * m_ms / 3, 1), "algorithmic_f32_tflops": round((m_fl / 1e12) / (m_ms / 1e3), 2) if m_ms > 0 else None,
                               "executed": "2*P*N1*N2*64 FLOP on v_mfma_f32_32x32x16_f16 (one sweep, every 32 x 32 tile multiplied once) + 32 exact fp32 similarities per flagged block",
                               # (m_fl = the algorithmic FLOPs of the 3 steps of the side pass; spans_us = us per ONE step)
                               "executed_f16_tflops_sweep": round(((m_fl / 3) / 1e12) / (spans_us.get(222, 0) / 1e6), 1) if spans_us.get(222) else None, "peak_f16_tflops": PEAK_F16_TFLOPS,
                               # the matcher's algorithmic floor: ONE fp16 GEMM per pair (2*N1*N2*64 FLOP) at the 2.5 PF peak, against everything xfh_match_mnn launches
                               "floor_algorithmic_us": round((m_fl / 3) / PEAK_F16_TFLOPS / 1e6, 1),
                               "frac_algorithmic": round(((m_fl / 3) / PEAK_F16_TFLOPS / 1e6) / (1e3 * m_ms / 3), 3) if m_ms > 0 else None},
            # the two 24 -> 24 convolutions on the fp16 matrix cores (fp16-pair arithmetic, fp32-equivalent results).  "achieved" prices the ALGORITHMIC fp32 work
            # against the f32 MFMA peak (comparable across rounds), "executed_f16_tflops" the three MFMAs per product (+ channel padding) against the fp16 peak.
            "roofline_conv24": {"bound": "mfma", "kernel": "conv_bxd_kernel<24,24> (block2.0 / block2.1 on v_mfma_f32_32x32x16_f16, three MFMAs per K = 16: fp16-pair arithmetic; the next tile staged inside this tile's MFMAs)",
                                "us_per_step": round(1e3 * b_ms / 3, 1), "launches_per_step": 2,
                                "achieved": round((b_fl / 1e12) / (b_ms / 1e3), 2) if b_ms > 0 else None, "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                                "executed_f16_tflops": round((b_fl * 3 * (32 / 24) * (224 / 216) / 1e12) / (b_ms / 1e3), 1) if b_ms > 0 and opt_fx.value & 2 else None,
                                "peak_f16_tflops": 2500.0},
            # Whole path.  frac: against the roofline of the INSTRUCTION MIX THAT SHIPS -- per kernel max(algorithmic bytes / 8 TB/s, executed FLOPs /
            # the peak of the pipe the kernel uses), summed (roofline_kernels) -- i.e. how close the kernels are to their own floors.
            # frac_vs_survey_roof: SURVEY 8(d)'s figure (direct fp32 convolutions at 157.3 TF + one f32 GEMM for the match = 26.9 us per frame): a
            # bound the implementation has left behind (the fp16-pair kernels run on a 16x faster pipe at three MFMAs per product).
            # hbm_fraction: the north_star's bar (>= 0.8 of the HBM roofline on 78.6 MB per frame) -- NOT met: the path is compute-side bound.
            # frac_timed: the same sum of floors against the WALL time of a timed step (all lanes, host included): what the chip delivers per step.
            "roofline_path": {"frac": k_sum["frac"], "sum_of_kernel_floors_us_per_step": k_sum["sum_of_floors_us_per_step"], "kernels_us_per_step": k_sum["kernels_us_per_step"],
                              "frac_timed": round(k_sum["sum_of_floors_us_per_step"] / (1e6 * dt_max / args.steps), 4),
                              "frac_vs_survey_roof": round(fps_gpu * T_ROOF_US_PER_FRAME / 1e6, 4), "t_roof_survey_us_per_frame": T_ROOF_US_PER_FRAME,
                              "hbm_fraction": round(ALGO_BYTES_PER_FRAME * fps_gpu / 8.0e12, 4), "north_star_hbm_bar": 0.8, "north_star_hbm_bar_met": bool(ALGO_BYTES_PER_FRAME * fps_gpu / 8.0e12 >= 0.8),
                              "algorithmic_bytes_per_frame": ALGO_BYTES_PER_FRAME, "algorithmic_flops_per_frame": ALGO_FLOPS_PER_FRAME,
                              "note": "per GPU; 78.6 MB and 2.622 + 1.074 GFLOP per frame (SURVEY 8d); at 8 TB/s the pure-HBM line is 102 k frames/s"},
            "roofline_kernels": k_rows,
            # every MFMA convolution launch behind block1 (12 per step), priced like conv_family_roofline: the FLOPs the fp16-pair kernels EXECUTE (three fp16 MFMAs per
            # product) against the fp16 MFMA peak, next to the algorithmic fp32 rate
            "roofline_conv_family": {"bound": "mfma", "kernel": "conv_bxd_kernel<24,24> x2, conv_bxs2_kernel<24>, conv_rs64_kernel (7 launches: 64 -> 64 and 128 -> 128 3x3, with and "
                                                                  "without the fused 1x1), conv_bx64s2x_kernel x2, conv_mfma_kernel<128,64,1x1> -- fp16-pair arithmetic but for the 1x1",
                                     "achieved": round(3 * (cfl / 1e12) / (cms / 1e3), 2) if cms > 0 else None, "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                                     "frac": round(3 * (cfl / 1e12) / (cms / 1e3) / PEAK_F16_TFLOPS, 4) if cms > 0 else None,
                                     "algorithmic_fp32_tflops": round((cfl / 1e12) / (cms / 1e3), 2) if cms > 0 else None,
                                     "us_per_step": round(1e3 * cms / 3, 1)},
        }
        if rep_fps:
            out["median_of_5x20_steps_fps"] = round(float(np.median(rep_fps)), 1)
            out["fps_of_5_windows"] = [round(v, 1) for v in rep_fps]
        out.update(side)
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        # The contract line stays short enough for an 8 KB tail: the per-kernel table and the long notes go out FIRST, on a line of their own that is not JSON by itself
        # ("# detail: {...}") on STDERR; stdout carries the ONE JSON line.
        detail = {k: out.pop(k) for k in ("roofline_kernels", "roofline_conv_family", "roofline_conv24", "side_workloads") if k in out}
        print("# detail: " + json.dumps(detail), file=sys.stderr, flush=True)      # (stderr: stdout carries exactly one line, the contract's JSON line)
        print(json.dumps(out))
    close_group(dist)


if __name__ == "__main__":
    main()
