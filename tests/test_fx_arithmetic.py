"""CPU restatement of the fp16-pair arithmetic of the split-operand convolutions (csrc/bx_split.hpp, weight_split.hpp: split_weight):
the identities and error bounds the kernels rely on, checked in numpy -- representation error of the pair, exactness of the partial products in an fp32
accumulator, the error of a K = 576 product sum against fp64 next to an fp32 fma chain and the bf16 three-way split, the range limits the library guards.
The constants are parsed from the sources, so a change there has to pass here."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPLIT = open(os.path.join(ROOT, "accelerated_features_amd", "csrc", "bx_split.hpp")).read()
API = open(os.path.join(ROOT, "accelerated_features_amd", "csrc", "api.hip")).read() + open(os.path.join(ROOT, "accelerated_features_amd", "csrc", "weight_split.hpp")).read()


def pair(v):
    """x -> (xh, xl) as fp32 values of the two fp16 numbers: xh = fp16(x), xl = fp16(2^11 (x - xh))  (split2_f16)."""
    v = np.asarray(v, np.float32)
    h = v.astype(np.float16).astype(np.float32)
    l = ((v - h) * np.float32(2048.0)).astype(np.float16).astype(np.float32)
    return h, l


def weight_triple(w):
    """w -> (q0, q1, q2): q0 = fp16(2^11 w), q1 = fp16(w), q2 = fp16(2^11 w - q0)  (weight_split.hpp: split_weight)."""
    w = np.asarray(w, np.float32)
    s = w * np.float32(2048.0)
    q0 = s.astype(np.float16).astype(np.float32)
    return q0, w.astype(np.float16).astype(np.float32), (s - q0).astype(np.float16).astype(np.float32)


def test_constants_in_the_sources_are_the_ones_restated_here():
    assert re.search(r"FX_SCALE_INV = 1\.f / 2048\.f", SPLIT) and re.search(r"\* 2048\.f", SPLIT)          # scale 2^11 on both sides
    assert re.search(r"FX_MAX_INPUT = 65504\.f", SPLIT)                                                       # the fp16 maximum: the range guard's threshold
    assert re.search(r"kFxMaxWeight = 31\.f", API) and re.search(r"v \* 2048\.f", API)
    assert 31.0 * 2048.0 < 65504.0                                                                            # a weight below the limit has a finite q0


def test_pair_represents_fp32_to_22_bits_and_never_loses_small_values():
    rs = np.random.RandomState(0)
    for scale in (1e-6, 1e-3, 1.0, 50.0, 3.0e4):
        x = np.clip(rs.randn(200000) * scale, -65000.0, 65000.0).astype(np.float32)      # (beyond 65504 the kernels report: test_gpu_parity's range test)
        h, l = pair(x)
        rec = h.astype(np.float64) + l.astype(np.float64) / 2048.0
        err = np.abs(rec - x.astype(np.float64))
        # 2^-22 relative where the high part is a normal fp16 number; below 2^-14 the two parts' subnormal spacing (2^-24, 2^-35) bounds the error absolutely
        assert np.all(err <= np.maximum(2.0 ** -22 * np.abs(x), 2.0 ** -35)), scale
    # the residual, scaled, is exact in fp32 and at most half the size of x: it overflows only where xh does
    x = np.float32(65000.0)
    h, l = pair(x)
    assert np.isfinite(h) and np.isfinite(l) and abs(l) <= 0.5 * abs(x) + 1


def test_weight_triple_carries_22_bits_of_the_scaled_weight():
    rs = np.random.RandomState(1)
    w = (rs.randn(100000) * 0.3).astype(np.float32)
    w[:10] = [30.9, -30.9, 1e-7, -1e-7, 6.1e-5, 0.0, 1.0, -1.0, 0.124999, 15.99]
    q0, q1, q2 = weight_triple(w)
    assert np.all(np.isfinite(q0)) and np.all(np.isfinite(q2))
    hi = (q0.astype(np.float64) + q2.astype(np.float64)) / 2048.0
    assert np.all(np.abs(hi - w.astype(np.float64)) <= np.maximum(2.0 ** -22 * np.abs(w), 2.0 ** -36))
    assert np.all(np.abs(q1 - w) <= np.maximum(2.0 ** -11 * np.abs(w), 2.0 ** -25))                            # q1 only meets the 2^-11-sized low part of x


def _acc(acc, a, b):
    """one MFMA: exact products of a K = 16 slice summed in wide precision, the accumulator rounded to fp32 once"""
    return (acc.astype(np.float64) + a.astype(np.float64) @ b.astype(np.float64)).astype(np.float32)


def test_three_products_give_an_fp32_accurate_sum_more_accurate_than_six_bf16_ones():
    rs = np.random.RandomState(2)
    K, N, M = 576, 2048, 32
    x = np.maximum(rs.randn(K, N), 0).astype(np.float32) * 3
    w = (rs.randn(M, K) / np.sqrt(K)).astype(np.float32)
    ref = w.astype(np.float64) @ x.astype(np.float64)
    sc = np.abs(ref).max()
    # fp32 fma chain
    chain = np.zeros((M, N), np.float32)
    for k in range(K):
        chain = (chain.astype(np.float64) + w[:, k:k + 1].astype(np.float64) * x[k:k + 1].astype(np.float64)).astype(np.float32)
    # fp16 pair: (q2, xh) (q1, xl) (q0, xh) per K = 16 step, one accumulator at scale 2^11
    xh, xl = pair(x)
    q0, q1, q2 = weight_triple(w)
    acc = np.zeros((M, N), np.float32)
    for k in range(0, K, 16):
        s = slice(k, k + 16)
        for a, b in ((q2, xh), (q1, xl), (q0, xh)):
            # every partial product is exact in fp32: 11 x 11 significant bits
            p = a[:, s][:, :, None].astype(np.float64) * b[s][None, :, :1].astype(np.float64)
            assert np.all(p == p.astype(np.float32))
            acc = _acc(acc, a[:, s], b[s])
    y = acc * np.float32(1.0 / 2048.0)
    # bf16 three-way split, six products per step
    def bf16(v):
        u = v.astype(np.float32).view(np.uint32)
        return ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).view(np.float32)
    def split3(v):
        h = bf16(v); m = bf16(v - h); return h, m, bf16(v - h - m)
    wh, wm, wl = split3(w); bh, bm, bl = split3(x)
    acc6 = np.zeros((M, N), np.float32)
    for k in range(0, K, 16):
        s = slice(k, k + 16)
        for a, b in ((wl, bh), (wh, bl), (wm, bm), (wm, bh), (wh, bm), (wh, bh)):
            acc6 = _acc(acc6, a[:, s], b[s])
    e_chain, e_pair, e_bf = (np.abs(v - ref).max() / sc for v in (chain, y, acc6))
    print(f"max |err| / max |y|: fp32 fma chain {e_chain:.3g}  fp16 pair (3 MFMAs) {e_pair:.3g}  bf16 x3 (6 MFMAs) {e_bf:.3g}")
    assert e_pair <= e_chain and e_pair <= e_bf and e_pair < 5e-7


def test_python_option_table_mirrors_the_library():
    """XFeatModel.OPTION_VALUES / DEFAULT_FX (host-side validation before a handle exists) against api.hip's option table, include/xfeat_hip.h's XFH_FX_* and kernels.hpp's defaults."""
    py = open(os.path.join(ROOT, "accelerated_features_amd", "xfeat.py")).read()
    keys_py = set(re.findall(r'"(\w+)": lambda', re.search(r"OPTION_VALUES = \{(.*?)\}\n", py).group(1)))
    keys_lib = set(re.findall(r'\{"(\w+)", &Options::\w+\}', API))
    assert keys_py == keys_lib == {"match_exact", "match_sweep", "block1", "fx", "resize2"}
    hdr = open(os.path.join(ROOT, "include", "xfeat_hip.h")).read()
    bits = {k: int(v) for k, v in re.findall(r"XFH_FX_(CONV64|CONV24|HEADS|FINE) = (\d+)", hdr)}
    all_bits = bits["CONV64"] | bits["CONV24"] | bits["HEADS"] | bits["FINE"]
    m = re.search(r"FX_CONV64, FX_CONV24, FX_HEADS, FX_FINE = (\d+), (\d+), (\d+), (\d+)", py)
    assert [int(v) for v in m.groups()] == [bits["CONV64"], bits["CONV24"], bits["HEADS"], bits["FINE"]]
    hpp = open(os.path.join(ROOT, "accelerated_features_amd", "csrc", "kernels.hpp")).read()
    assert eval(re.search(r"int fx = ([\d |]+);", hpp).group(1)) == all_bits
    assert int(re.search(r"int block1 = (\d+);", hpp).group(1)) == int(re.search(r"DEFAULT_BLOCK1 = (\d+)", py).group(1)) == 7
    from accelerated_features_amd.xfeat import XFeatModel, DEFAULT_FX
    assert DEFAULT_FX == all_bits
    ok = XFeatModel.OPTION_VALUES
    assert ok["fx"](0) and ok["fx"](all_bits) and ok["fx"](bits["HEADS"]) and not ok["fx"](4) and not ok["fx"](all_bits | 128) and not ok["fx"](-1)
    assert ok["block1"](5) and ok["block1"](7) and not ok["block1"](0) and not ok["block1"](6)
    assert "heads_f32" not in ok and "wino" not in ok and "bx" not in ok
