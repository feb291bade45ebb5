"""CPU-side checks: the C-ABI library loads and exports every declared symbol, the Python
surface mirrors the reference's, host-side planning functions work without a GPU, and the
product path refuses to run without one (no fallback)."""
import inspect
import os
import re

import numpy as np
import pytest
import torch

import fixtures

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from accelerated_features_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from accelerated_features_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "xfeat_hip.h")).read()
    declared = set(re.findall(r"\b(xfh_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 20
    for s in declared:
        assert hasattr(lib, s), f"{s} declared but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes prototype"
    assert set(_lib.SIGNATURES) <= declared
    assert lib.xfh_version() == int(re.search(r"#define XFH_VERSION (\d+)", hdr).group(1))


def test_weight_table_matches_spec_and_reference_keys(lib):
    from accelerated_features_amd import XFeat
    from accelerated_features_amd.spec import state_dict_keys
    sd = fixtures.synthetic_state_dict(0)
    xf = XFeat(weights=sd)
    mine = xf.net.state_dict()
    assert list(mine.keys()) == list(sd.keys()) or set(mine.keys()) == set(sd.keys())
    assert len(mine) == 122                                   # key count of the reference XFeatModel
    for k, shape in state_dict_keys().items():
        assert tuple(mine[k].shape) == tuple(shape), k
        assert torch.equal(mine[k], sd[k]), k
    arrs = xf.net.weight_arrays()
    assert len(arrs) == lib.xfh_num_weight_arrays() == 95
    for i, a in enumerate(arrs):
        assert a.dtype == np.float32 and a.size == lib.xfh_weight_array_floats(i)
    assert sum(a.size for a in arrs if True) > 1_544_758      # params + BN statistics


def test_public_surface_mirrors_reference():
    """Names, argument names and defaults of modules/xfeat.py::XFeat."""
    from accelerated_features_amd import XFeat
    expect = {
        "detectAndCompute": ["x", "top_k", "detection_threshold"],
        "detectAndComputeDense": ["x", "top_k", "multiscale"],
        "match_lighterglue": ["d0", "d1", "min_conf"],
        "match_xfeat": ["img1", "img2", "top_k", "min_cossim"],
        "match_xfeat_star": ["im_set1", "im_set2", "top_k"],
        "preprocess_tensor": ["x"],
        "get_kpts_heatmap": ["kpts", "softmax_temp"],
        "NMS": ["x", "threshold", "kernel_size"],
        "batch_match": ["feats1", "feats2", "min_cossim"],
        "subpix_softmax2d": ["heatmaps", "temp"],
        "refine_matches": ["d0", "d1", "matches", "batch_idx", "fine_conf"],
        "match": ["feats1", "feats2", "min_cossim"],
        "create_xy": ["h", "w", "dev"],
        "extract_dualscale": ["x", "top_k", "s1", "s2"],
        "parse_input": ["x"],
    }
    for name, args in expect.items():
        sig = inspect.signature(getattr(XFeat, name))
        assert list(sig.parameters)[1:1 + len(args)] == args, (name, list(sig.parameters))
    d = {n: p.default for n, p in inspect.signature(XFeat.__init__).parameters.items()}
    assert d["top_k"] == 4096 and d["detection_threshold"] == 0.05 and d["weights"].endswith("/../weights/xfeat.pt")
    assert inspect.signature(XFeat.match).parameters["min_cossim"].default == 0.82
    assert inspect.signature(XFeat.match_xfeat).parameters["min_cossim"].default == -1
    assert inspect.signature(XFeat.refine_matches).parameters["fine_conf"].default == 0.25
    assert inspect.signature(XFeat.extract_dualscale).parameters["s1"].default == 0.6
    assert inspect.signature(XFeat.extractDense).parameters["top_k"].default == 8_000
    import hubconf
    hs = inspect.signature(hubconf.XFeat).parameters
    assert [hs[k].default for k in ("pretrained", "top_k", "detection_threshold")] == [True, 4096, 0.05]


def test_host_helpers_match_reference_semantics():
    from accelerated_features_amd import XFeat
    xf = XFeat(weights=None)
    img = (np.random.RandomState(0).rand(20, 30, 3) * 255).astype(np.uint8)
    t = xf.parse_input(img)
    assert t.shape == (1, 3, 20, 30) and t.dtype == torch.float32 and float(t.max()) <= 1.0
    assert xf.parse_input(torch.zeros(3, 8, 8)).shape == (1, 3, 8, 8)
    xy = xf.create_xy(2, 3, "cpu")
    assert xy.tolist() == [[0, 0], [1, 0], [2, 0], [0, 1], [1, 1], [2, 1]]
    o = torch.full((1, 8, 8), -50.0)
    o[0, 2, 5] = 50.0
    assert torch.allclose(xf.subpix_softmax2d(o), torch.tensor([[1.0, -2.0]]))
    assert xf.top_k == 4096 and xf.detection_threshold == 0.05 and hasattr(xf.net, "fine_matcher")


def test_workspace_planning_is_host_only(lib):
    a = lib.xfh_backbone_workspace_bytes(64, 3, 480, 640)
    b = lib.xfh_backbone_workspace_bytes(1, 3, 480, 640)
    assert a > 60 * b > 0 and a % 256 == 0
    assert lib.xfh_backbone_workspace_bytes(0, 3, 480, 640) == 0
    assert lib.xfh_detect_workspace_bytes(64, 480, 640, 4096, 38400) > 64 * 38400 * 12
    assert lib.xfh_match_workspace_bytes(32, 4096, 4096) >= 32 * 4096 * (8 + 4 + 4)     # column keys + row arg-max + row max
    assert lib.xfh_refine_workspace_bytes(2, 511) > 2 * 511 * 512 * 4 * 2
    assert lib.xfh_dense_workspace_bytes(2, 20, 24, 100) > 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_no_gpu_means_loud_failure_not_fallback(lib):
    from accelerated_features_amd import XFeat, _lib
    xf = XFeat(weights=fixtures.synthetic_state_dict(0))
    with pytest.raises(_lib.XFeatHipError):
        xf.detectAndCompute(torch.rand(1, 3, 64, 64))
    with pytest.raises(_lib.XFeatHipError):
        xf.match(torch.rand(10, 64), torch.rand(10, 64))
    with pytest.raises(_lib.XFeatHipError):
        xf.match_xfeat_star(torch.rand(2, 3, 64, 64), torch.rand(2, 3, 64, 64))
    with pytest.raises(_lib.XFeatHipError):
        xf.net(torch.rand(1, 3, 64, 64))
    with pytest.raises(RuntimeError):
        xf.match_lighterglue({}, {})
    from accelerated_features_amd.streaming import FrameStream
    with pytest.raises(_lib.XFeatHipError):
        FrameStream(weights=fixtures.synthetic_state_dict(0), lanes=2)
    # the C ABI itself refuses to create a context without a device
    import ctypes as C
    arrs = xf.net.weight_arrays()
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    h = C.c_void_p()
    rc = lib.xfh_create(ptrs, len(arrs), 0, C.byref(h))
    assert rc != 0 and b"device" in lib.xfh_last_error().lower()
    assert lib.xfh_create(ptrs, 3, 0, C.byref(h)) == -2       # malformed weight table


def test_range_fallback_lands_on_fp32_range_kernels(monkeypatch):
    """XFeatModel.fx_range_exceeded (the reaction to bit 0 of the status word: an activation beyond fp16's range) must end on kernels with fp32's range -- fx = 0 (f32-MFMA
    convolutions, heads, linear layers), block1 = 5 (vector ALUs): include/xfeat_hip.h "THE RANGE FALLBACK" -- and must not ask for a second repeat; checked for the shipped
    defaults and for partial option sets, without a GPU (options are remembered until a handle exists)."""
    import warnings
    from accelerated_features_amd import XFeat, _lib, xfeat as xm
    for overrides in ({}, {"fx": xm.FX_HEADS}, {"fx": 0}, {"block1": 5}, {"fx": xm.FX_CONV64 | xm.FX_FINE, "block1": 5}):
        net = XFeat(weights=None).net
        for k, v in overrides.items():
            net.set_option(k, v)
        assert net.fx_range_exceeded(status=0) is False and net.fx_range_exceeded(status=2) is False
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            assert net.fx_range_exceeded(status=1) is True
        assert any("fp16-pair" in str(m.message) for m in w)
        assert net._effective_option("fx") == 0 and net._effective_option("block1") == 5
        assert net.fx_range_exceeded(status=1) is False                          # nothing left to switch off: no endless repeat on a stale flag
    net = XFeat(weights=None).net
    net.set_option("fx", 0); net.set_option("block1", 5)
    assert net.fx_range_exceeded() is False                                      # fp32-range kernels only: no read-back at all
    for key, value in (("heads_f32", 0), ("wino", 1), ("bx", 21), ("block1", 0), ("block1", 6), ("fx", 4), ("fx", 3979), ("fx", 4095)):      # what round 6 deleted is refused
        with pytest.raises(_lib.XFeatHipError):
            net.set_option(key, value)


def test_python_mirrors_of_the_library_defaults_agree_with_kernels_hpp():
    """xfeat.DEFAULT_FX / DEFAULT_BLOCK1 mirror csrc/kernels.hpp: Options (the GPU suite asks the live handle; this is the same check without one)."""
    import re
    from accelerated_features_amd import xfeat as xm
    src = open(os.path.join(os.path.dirname(xm.__file__), "csrc", "kernels.hpp"), encoding="utf-8").read()
    body = src[src.index("struct Options {"):]
    body = body[:body.index("};")]
    got = {k: eval(v) for k, v in re.findall(r"^\s*int\s+(\w+)\s*=\s*([\d |]+)\s*;", body, flags=re.M)}
    assert (got["fx"], got["block1"]) == (xm.DEFAULT_FX, xm.DEFAULT_BLOCK1) and set(got) == {"match_exact", "match_sweep", "fx", "resize2", "block1"}, got


def test_the_library_holds_no_retired_kernel():
    """one default and one fp32-range fallback per layer.  No bf16 matrix instruction, no Winograd / conv_bx64 / head_fused kernel and no debug soak entry in the
    built library; the header does not describe anything as unsafe."""
    import subprocess
    from accelerated_features_amd import build
    assert os.path.exists(build.LIB), "build the library first (python -m accelerated_features_amd.build)"
    syms = subprocess.run(["nm", "-D", "--defined-only", build.LIB], check=True, capture_output=True, text=True).stdout
    assert "xfh_debug_head_soak" not in syms and "xfh_set_option" in syms
    names = subprocess.run(["strings", build.LIB], check=True, capture_output=True, text=True).stdout
    for gone in ("conv_wino_kernel", "conv_bx64_kernel", "conv_bx64s2_kernel", "head_fused_kernel", "mfma_f32_32x32x16_bf16", "mfma_f32_16x16x32_bf16"):
        assert gone not in names, gone
    for kept in ("conv_rs64_kernel", "conv_bx64s2x_kernel", "conv_bx_kernel", "head_bx_kernel", "head_f32r_kernel", "block1_mx_kernel", "block1_fused_kernel", "conv_mfma_kernel"):
        assert kept in names, kept
    hdr = open(os.path.join(ROOT, "include", "xfeat_hip.h")).read()
    assert "NOT safe" not in hdr and "heads_f32" not in hdr and "wino" not in hdr.lower()
    import glob
    for f in glob.glob(os.path.join(ROOT, "accelerated_features_amd", "csrc", "*")):
        assert "_bf16(" not in open(f).read() or os.path.basename(f) in ("k_lighterglue.hip", "api_lg.hip"), f



def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "accelerated_features_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", ""), f"{f} mentions the oracle"
                assert "/root/reference" not in src or f.endswith(".py") and "import" not in src.split("/root/reference")[0][-40:], f


def test_every_barrier_in_dma_kernels_waits_for_the_dma():
    """ISA audit (tools/check_dma_barriers.py): hipcc was seen dropping the vmcnt(0) in front of a
    loop-back-edge s_barrier in a global->LDS DMA kernel (24->24 conv): ~7 % of launches were wrong.
    Every barrier of every DMA kernel must be preceded by an explicit vmcnt(0) wait."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_dma_barriers", os.path.join(ROOT, "tools", "check_dma_barriers.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import glob
    def _src_dma(f):      # the file and the kernel-body headers it includes (csrc/*_body.hpp)
        t = open(f).read()
        return t + "".join(open(os.path.join(os.path.dirname(f), h)).read() for h in re.findall(r'#include "(\w+_body\.hpp)"', t))
    files = [f for f in sorted(glob.glob(os.path.join(ROOT, "accelerated_features_amd", "csrc", "*.hip")))
             if re.search(r"global_load_lds|buffer_load[^\n]* lds", _src_dma(f))]
    assert len(files) >= 3
    mod.isa_asm.prefetch(sorted(glob.glob(os.path.join(ROOT, "accelerated_features_amd", "csrc", "k_*.hip"))))      # (every kernel file once, in parallel; the audit below shares them)
    for f in files:
        nk, nb, bad = mod.audit(f)
        assert nk > 0 and nb > 0
        assert not bad, (os.path.basename(f), bad[:3])


def test_no_valu_write_lands_in_a_freshly_read_bf16_mfma_operand():
    """ISA audit (tools/check_mfma_war.py): a VALU result written a few cycles after a K = 16 bf16 MFMA was issued can land in A / B lanes
    the matrix core has not read yet (split-bf16 heads: wrong logits for lanes 16-31 of a wave in a few launches).  No VALU instruction
    may write a register that one of the preceding bf16 MFMAs (8 instructions) reads as A or B."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_mfma_war", os.path.join(ROOT, "tools", "check_mfma_war.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import glob
    def _src(f):      # the file and the kernel-body headers it includes (csrc/*_body.hpp)
        t = open(f).read()
        return t + "".join(open(os.path.join(os.path.dirname(f), h)).read() for h in re.findall(r'#include "(\w+_body\.hpp)"', t))
    files = [f for f in sorted(glob.glob(os.path.join(ROOT, "accelerated_features_amd", "csrc", "*.hip"))) if re.search(r"mfma_f32_\d+x\d+x\d+_(bf16|f16)\(", _src(f))]
    assert len(files) >= 4
    for f in files:
        nk, nm, bad = mod.audit(f)
        assert nk > 0 and nm > 0
        assert not bad, (os.path.basename(f), bad[:3])


def test_no_kernel_in_the_library_uses_scratch(tmp_path):
    """Code-object metadata of the built libxfeat_hip.so: no kernel has a private segment (scratch) or spills vector registers
    (a scratch reload parks a vmcnt(0) wherever it lands; the variants that needed scratch were measured slower and removed)."""
    import shutil
    import subprocess
    import yaml
    from accelerated_features_amd import build
    lib = build.LIB
    assert os.path.exists(lib), "build the library first (python -m accelerated_features_amd.build)"
    llvm = "/opt/rocm/lib/llvm/bin"
    so = shutil.copy(lib, tmp_path / "lib.so")
    subprocess.run([f"{llvm}/llvm-objdump", "--offloading", os.path.basename(so)], cwd=tmp_path, check=True, capture_output=True)
    objs = sorted(p for p in os.listdir(tmp_path) if p.endswith("gfx950"))
    assert objs, "no gfx950 code objects in the library"
    n = 0
    for o in objs:
        notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", o], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        if "amdhsa.kernels" not in notes:
            continue
        doc = notes[notes.index("---"):notes.rindex("...")]
        for k in yaml.safe_load(doc)["amdhsa.kernels"]:
            n += 1
            assert k[".private_segment_fixed_size"] == 0 and k[".vgpr_spill_count"] == 0, (k[".name"], k[".private_segment_fixed_size"], k[".vgpr_spill_count"])
    assert n >= 100


def test_find_homography_has_cv2s_signature_and_no_cpu_path():
    """cv2.findHomography(srcPoints, dstPoints[, method[, ransacReprojThreshold[, mask[, maxIters[, confidence]]]]]) -- the call of
    realtime_demo.py:225 binds positionally / by these keywords; without a GPU the function raises (there is no host estimator in the product)."""
    import inspect
    import numpy as np
    import torch
    from accelerated_features_amd import _lib, homography
    sig = inspect.signature(homography.find_homography)
    names = list(sig.parameters)
    assert names[:7] == ["srcPoints", "dstPoints", "method", "ransacReprojThreshold", "mask", "maxIters", "confidence"]
    assert sig.parameters["ransacReprojThreshold"].default == 3.0 and sig.parameters["maxIters"].default == 2000 and sig.parameters["confidence"].default == 0.995
    assert homography.USAC_MAGSAC == 38                         # cv2.USAC_MAGSAC
    sig.bind(np.zeros((8, 2)), np.zeros((8, 2)), 38, 4.0, maxIters=700, confidence=0.995)
    if not torch.cuda.is_available():
        with pytest.raises(_lib.XFeatHipError):
            homography.find_homography(np.zeros((8, 2)), np.zeros((8, 2)), 38, 4.0, maxIters=700, confidence=0.995)
        with pytest.raises(_lib.XFeatHipError):
            homography.find_homography_batch(torch.zeros(1, 8, 2), torch.zeros(1, 8, 2))


def _build_c_host(tmp_path):
    import subprocess
    from accelerated_features_amd import build
    exe = str(tmp_path / "c_host_homography")
    libdir = os.path.dirname(build.LIB)
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "examples", "c_host_homography.c"), "-L", libdir, "-lxfeat_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_plain_c_and_a_c_host_links(tmp_path):
    """include/xfeat_hip.h is the boundary for hosts in any language: it must compile as strict C99 and as C++11 on its own, and a C program
    that uses only it and the HIP runtime must link against libxfeat_hip.so (examples/c_host_homography.c; the GPU suite runs it)."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "xfeat_hip.h"\nint main(void) { return xfh_version() >= 100 ? 0 : 1; }\n')
    for cc, std in (("gcc", ["-std=c99", "-pedantic"]), ("g++", ["-std=c++11", "-x", "c++"])):
        r = subprocess.run([cc, *std, "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "t.o")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    assert os.path.exists(_build_c_host(tmp_path))


@pytest.mark.gpu
def test_c_host_runs_the_homography_stage(tmp_path):
    import subprocess
    r = subprocess.run([_build_c_host(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert "mask differences 0" in r.stdout
