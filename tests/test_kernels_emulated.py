"""Shipped kernels compiled for the HOST and run against float64 references without a GPU.  (The bodies of block1, the heads, conv_rs64, the stride-2 64-channel kernel
and the fine_matcher's layers live in csrc/*_body.hpp and have tests of their own; what is sliced here are kernels of files that are still self-contained: k_conv_bx.hip,
k_preproc.hip, k_match_f16.hip.)  The kernel source is SLICED out of the product files -- nothing in them is
changed for this -- and compiled with the host clang against tests/emu/emu.hpp: one host thread per work-item, LDS as a buffer (initialised to NaN patterns),
__syncthreads a barrier, the LDS-DMA a copy, the matrix instructions and the lane exchanges emulated.  What it checks: index arithmetic, tile and weight layouts,
partial tiles, the barrier structure; what it cannot: timing, memory ordering, hardware hazards (the GPU suite and the soaks do that).  The slicing is by markers
in the source: a change there that moves them fails this test loudly instead of silently testing something else."""
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "accelerated_features_amd", "csrc")
EMU = os.path.join(ROOT, "tests", "emu")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _between(text, start, end):
    a = text.index(start)
    return text[a:text.index(end, a)]


def _must_sub(text, old, new):
    assert old in text, f"marker not found in the kernel source: {old[:70]!r}"
    return text.replace(old, new)


def _slice_pyramid():
    """pyramid_sum_kernel (x3 + up(x4) + up(x5): source planes and per-plane coefficient tables in LDS) with the interpolation helpers it shares with the resize kernels"""
    t = open(os.path.join(CSRC, "k_preproc.hip")).read()
    s = _between(t, "__device__ inline void lin_coef(", "__global__ __launch_bounds__(256) void resize_bilinear_kernel(")
    s += _between(t, "__device__ inline float bilerp_at(", "constexpr int PYR53_CG = ")      # pyramid_sum_kernel and pyramid53_kernel (block5.3 fused in)
    s = _must_sub(s, "__global__ __launch_bounds__(256) void pyramid_sum_kernel(", "inline void pyramid_sum_kernel(")
    s = _must_sub(s, "__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))", "inline")
    s = _must_sub(s, "extern __shared__ __attribute__((aligned(16))) float sm[];", "XFH_DYN_LDS(sm);")
    assert "asm volatile" not in s and "<<<" not in s
    return s


def _slice_gray():
    """gray_stats_kernel<CT> (channel mean + fp64 partial sums per chunk) and gray_coef_kernel (the instance normalisation as {alpha, beta} per image): the 2-D grid becomes a
    1-D one, the static LDS array a pointer into the emulator's LDS"""
    t = open(os.path.join(CSRC, "k_preproc.hip")).read()
    s = _between(t, "constexpr int GS_UNROLL = 5;", "// uint8 ingest (XFeat.parse_input")
    s += _between(t, "__global__ __launch_bounds__(64) void gray_coef_kernel(", "void launch_gray_norm(")
    s = _must_sub(s, "__global__ __launch_bounds__(256) void gray_stats_kernel(", "inline void gray_stats_kernel(")
    s = _must_sub(s, "__global__ __launch_bounds__(64) void gray_coef_kernel(", "inline void gray_coef_kernel(")
    s = _must_sub(s, "const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;", "const int b = blockIdx.x / GS_CHUNKS, ch = blockIdx.x % GS_CHUNKS, tid = threadIdx.x;")
    s = _must_sub(s, "__shared__ double sm[8];", "double* sm = reinterpret_cast<double*>(emu::wg->lds_base());")
    assert "asm volatile" not in s and "<<<" not in s and "blockIdx.y" not in s
    return s


def _slice_resize2():
    """both forms of the fused two-stage resize (lin_coef / bilerp, the R2 constants, resize2_gray_stats_kernel, resize2_gray_stats_lds_kernel) and gray_coef_kernel: the 2-D
    grid becomes a 1-D one, the dynamic LDS a pointer into the emulator's LDS, the static partial-sum array its last 64 bytes"""
    t = open(os.path.join(CSRC, "k_preproc.hip")).read()
    s = _between(t, "__device__ inline void lin_coef(", "__global__ __launch_bounds__(256) void resize_bilinear_kernel(")
    s += _between(t, "constexpr int R2_TW = 64,", "// returns -1 when the stage-2 step is too large for the LDS region")
    s += _between(t, "__global__ __launch_bounds__(64) void gray_coef_kernel(", "void launch_gray_norm(")
    s = _must_sub(s, "__global__ __launch_bounds__(256) void resize2_gray_stats_kernel(", "inline void resize2_gray_stats_kernel(")
    s = _must_sub(s, "__global__ __launch_bounds__(256, 4) void resize2_gray_stats_lds_kernel(", "inline void resize2_gray_stats_lds_kernel(")
    s = _must_sub(s, "__global__ __launch_bounds__(64) void gray_coef_kernel(", "inline void gray_coef_kernel(")
    s = _must_sub(s, "const int b = blockIdx.y, ch = blockIdx.x, tid = threadIdx.x;", "const int b = blockIdx.x / GS_CHUNKS, ch = blockIdx.x % GS_CHUNKS, tid = threadIdx.x;")
    s = _must_sub(s, "extern __shared__ float mid[];", "float* mid = reinterpret_cast<float*>(emu::wg->lds_base());")
    s = _must_sub(s, "extern __shared__ __attribute__((aligned(16))) float r2sm[];", "float* r2sm = reinterpret_cast<float*>(emu::wg->lds_base());")
    s = _must_sub(s, "__shared__ double sm[8];", "double* sm = reinterpret_cast<double*>(emu::wg->lds_base() + emu::wg->lds.size() - 128);")
    s = _must_sub(s, "__shared__ double sm2[8];", "double* sm2 = reinterpret_cast<double*>(emu::wg->lds_base() + emu::wg->lds.size() - 128);")
    assert "asm volatile" not in s and "<<<" not in s and "blockIdx.y" not in s
    return s


def _slice_match_sweep():
    """mnn_f16_sweep_kernel with its constants and the 16-value maximum tree: the three static LDS arrays become pointers into the emulator's LDS, the register keep-alive an
    empty statement"""
    t = open(os.path.join(CSRC, "k_match_f16.hip")).read()
    s = _between(t, "typedef float f32x16 __attribute__((ext_vector_type(16)));", "// 16 lanes per descriptor row (float4 each), 16 rows per pass")
    s += _between(t, "// Block maxima and thresholds travel as fp16", "// E in the units of the scaled product")
    s += _between(t, "constexpr int FT_TILES = FT_COLS / 32;", "// thresholds, once the maxima are complete")
    s = _must_sub(s, "__global__ __launch_bounds__(512) void mnn_f16_sweep_kernel(", "inline void mnn_f16_sweep_kernel(")
    s = _must_sub(s, "__global__ __launch_bounds__(64 * S2_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2)))\nvoid mnn_f16_sweep2_kernel(", "inline void mnn_f16_sweep2_kernel(")
    s = _must_sub(s, "__shared__ __attribute__((aligned(16))) _Float16 Dl2[S2_COLS * FT_DS];", "_Float16* Dl2 = reinterpret_cast<_Float16*>(emu::wg->lds_base());")
    s = _must_sub(s, "__shared__ __attribute__((aligned(16))) _Float16 af[S2_WAVES][64 * FT_DS];", "_Float16 (*af)[64 * FT_DS] = reinterpret_cast<_Float16 (*)[64 * FT_DS]>(emu::wg->lds_base() + sizeof(_Float16) * S2_COLS * FT_DS);")
    s = _must_sub(s, "__host__ __device__ inline int s2_c_blocks", "inline int s2_c_blocks")
    s = _must_sub(s, "    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);\n    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));",
                  "    return fmaxf(v, xhalf(v));")      # (the lane exchange by the emulator's shuffle)
    s = _must_sub(s, "__shared__ __attribute__((aligned(16))) _Float16 Dl[FT_COLS * FT_DS];", "_Float16* Dl = reinterpret_cast<_Float16*>(emu::wg->lds_base());")
    s = _must_sub(s, "__shared__ float colx[8][FT_COLS];", "float (*colx)[FT_COLS] = reinterpret_cast<float (*)[FT_COLS]>(emu::wg->lds_base() + sizeof(_Float16) * FT_COLS * FT_DS);")
    s = _must_sub(s, "__shared__ int next_block;", "int& next_block = *reinterpret_cast<int*>(emu::wg->lds_base() + sizeof(_Float16) * FT_COLS * FT_DS + sizeof(float) * 8 * FT_COLS);")
    s = _must_sub(s, "XFH_KEEP_FRAGS(bfrag[ct & 1]);", ";")
    s = _must_sub(s, "nxt = __builtin_amdgcn_readfirstlane(t);", "nxt = emu_bcast0(t);")      # (only lane 0 holds t: a real broadcast, emu.hpp's readfirstlane is the identity)
    n0 = s.count('asm volatile("" :')
    s = s.replace('asm volatile("" :', "XFH_EMU_NOASM(")      # value pins of the one-orientation sweep (empty statements with register constraints: nothing to run)
    assert n0 == 7 and "asm volatile" not in s and "<<<" not in s and "__shared__" not in s
    return s


def _slice_conv_bx24():
    """conv_bx_kernel<24, 24> (block2.0 / block2.1) and conv_bxs2_kernel<24> (block3.0): weights in registers, one staged halo tile per output tile"""
    t = open(os.path.join(CSRC, "k_conv_bx.hip")).read()
    s = _between(t, "struct BxArgs {", "template <int CIN, bool IN_CL>\nstatic int run_bxs2(")
    for name, args, nthr in (("conv_bx_kernel", "BxArgs", 256), ("conv_bxd_kernel", "BxArgs", 512), ("conv_bxs2_kernel", "BxS2Args", 256)):
        s = _must_sub(s, f"__global__ __launch_bounds__({nthr}) __attribute__((amdgpu_waves_per_eu(2, 2)))\nvoid {name}({args} a) {{", f"inline void {name}({args} a) {{")
    assert s.count("extern __shared__ __attribute__((aligned(16))) unsigned char smem_bx[];") == 3
    s = s.replace("extern __shared__ __attribute__((aligned(16))) unsigned char smem_bx[];", "XFH_DYN_LDS_BYTES(smem_bx);")
    n0 = s.count("asm volatile")
    s = _must_sub(s, 'asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(lds_alloc));', "lds_alloc = 0;")      # (which workgroup of a CU this is: a start delay, nothing else)
    s = re.sub(r'asm volatile\("s_nop 7[^;]*;', ";", s)                                   # idle slots tied to the accumulators
    s = re.sub(r'asm volatile\("" : "\+v"[^;]*;', ";", s)                                 # "the wait for the weight loads belongs here": register pins
    assert n0 == 9 and "asm volatile" not in s, "an inline-assembly statement of k_conv_bx.hip is not covered"
    assert "<<<" not in s
    return s


def _slice_weight_split():
    t = open(os.path.join(CSRC, "weight_split.hpp")).read()      # (this branch: the host-side split helpers have a header of their own)
    return _between(t, "// fp32 -> fp16, round to nearest even", "constexpr float kFxMaxWeight")


def _slice_bx_split():
    t = open(os.path.join(CSRC, "bx_split.hpp")).read()
    s = _between(t, "typedef float f32x16 __attribute__((ext_vector_type(16)));", "}  // namespace xfh")
    return s


@pytest.fixture(scope="module")
def emu_bins():
    if not os.path.exists(CLANG):
        pytest.skip("no host clang")
    td = tempfile.mkdtemp()
    open(os.path.join(td, "conv_bx24_slice.hpp"), "w").write(_slice_conv_bx24())
    open(os.path.join(td, "pyramid_slice.hpp"), "w").write(_slice_pyramid())
    open(os.path.join(td, "gray_slice.hpp"), "w").write(_slice_gray())
    open(os.path.join(td, "resize2_slice.hpp"), "w").write(_slice_resize2())
    open(os.path.join(td, "match_sweep_slice.hpp"), "w").write(_slice_match_sweep())
    open(os.path.join(td, "weight_split_slice.hpp"), "w").write(_slice_weight_split())
    open(os.path.join(td, "bx_split_slice.hpp"), "w").write(_slice_bx_split())
    out = {}
    for name in ("conv_bx24_emu", "pyramid_emu", "pyramid53_emu", "gray_emu", "resize2_emu", "match_sweep_emu"):
        out[name] = os.path.join(td, name)
        subprocess.run([CLANG, "-O1", "-w", "-std=c++20", "-pthread", "-I", td, "-I", EMU, os.path.join(EMU, name + ".cpp"), "-o", out[name]], check=True)
    return out


def _blob(hdr, arrs):
    return np.concatenate([np.array(hdr, np.int32).view(np.float32)] + [np.asarray(a, np.float32).reshape(-1) for a in arrs]).tobytes()


@pytest.mark.parametrize("stride,shape,grid", [(1, (1, 16, 64), 2), (1, (1, 8, 32), 1), (1, (2, 21, 45), 3), (2, (1, 16, 64), 2), (2, (1, 8, 32), 1), (2, (2, 21, 45), 3), (1, (8, 8, 32), 8),
                                               (1, (1, 48, 96), 2), (1, (3, 35, 70), 1), (1, (9, 16, 32), 8), (3, (2, 21, 45), 3), (3, (1, 16, 64), 2),
                                               (5, (2, 21, 45), 3), (5, (1, 48, 96), 2), (7, (2, 21, 45), 3), (7, (1, 48, 96), 2), (7, (9, 16, 32), 8), (8, (2, 21, 45), 3), (8, (1, 16, 64), 2)])
def test_conv_bx24_kernels_on_the_host(emu_bins, stride, shape, grid):
    """the 24-channel layers on the fp16 matrix cores with their weights in registers: conv_bxd_kernel<24, 24> (block2.0 / block2.1: 16 x 32 tiles, the next tile requested and
    staged inside this tile's MFMAs, two tile buffers; stride code 3 = conv_bx_kernel, the form the trace build keeps) and conv_bxs2_kernel<24> (block3.0, stride 2,
    64 couts) in the fp16-pair arithmetic: full tiles, partial tiles with odd sizes (21 x 45, 35 x 70), several tiles per workgroup (three and more: both buffers re-used), one
    workgroup for everything, more workgroups than tiles, the XCD mapping of the work list (B = 8 / 9 on a grid of 8); stride codes 5 / 7 / 8: the channels-last forms of the
    backbone's links block2.0 -> block2.1 -> block3.0 (16-byte loads of an item's 8 channels, 16-byte stores of a lane's 4 couts)"""
    B, H, W = shape
    cout = 64 if stride in (2, 8) else 24
    g = torch.Generator().manual_seed(7 * stride + 1 + H)
    x = torch.relu(torch.randn(B, 24, H, W, generator=g)) * 2
    w = torch.randn(cout, 24, 3, 3, generator=g) / 15
    b = torch.randn(cout, generator=g) * 0.3
    out = subprocess.run([emu_bins["conv_bx24_emu"]], input=_blob([B, H, W, stride, 1, grid], [x, w, b]), capture_output=True, check=True, timeout=400).stdout
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride={3: 1, 5: 1, 7: 1, 8: 2}.get(stride, stride), padding=1))
    y = np.frombuffer(out[:-4], np.float32).reshape(tuple(ref.shape))
    d = np.abs(y - ref.numpy())
    print(f"conv_bx24 stride {stride} {shape}: max |err| {d.max():.3g}, max |y| {float(ref.abs().max()):.3g}")
    assert int(np.frombuffer(out[-4:], np.int32)[0]) == 0 and np.isfinite(y).all() and d.max() <= 3e-6 * float(ref.abs().max())


@pytest.mark.parametrize("shape,use_lds", [((3, 12, 16), 1), ((2, 60, 80), 1), ((2, 9, 14), 1), ((2, 12, 16), 0), ((1, 10, 13), 0)])
def test_pyramid_sum_kernel_on_the_host(emu_bins, shape, use_lds):
    """x3 + up(x4) + up(x5) (modules/model.py:146-148: F.interpolate bilinear, align_corners=False): planes whose width is / is not a multiple of four (float4 / scalar path),
    the LDS-staged form with its coefficient tables and the direct-gather fall-back, against ATen in float64"""
    planes, H3, W3 = shape
    H4, W4, H5, W5 = (H3 + 1) // 2, (W3 + 1) // 2, (H3 + 3) // 4, (W3 + 3) // 4
    g = torch.Generator().manual_seed(H3 * W3)
    x3, x4, x5 = torch.randn(1, planes, H3, W3, generator=g), torch.randn(1, planes, H4, W4, generator=g), torch.randn(1, planes, H5, W5, generator=g)
    out = subprocess.run([emu_bins["pyramid_emu"]], input=_blob([planes, H3, W3, H4, W4, H5, W5, use_lds], [x3, x4, x5]), capture_output=True, check=True, timeout=240).stdout
    F = torch.nn.functional
    ref = x3.double() + F.interpolate(x4.double(), (H3, W3), mode="bilinear") + F.interpolate(x5.double(), (H3, W3), mode="bilinear")
    y = np.frombuffer(out, np.float32).reshape(1, planes, H3, W3)
    d = np.abs(y - ref.numpy())
    print(f"pyramid_sum {shape} lds {use_lds}: max |err| {d.max():.3g}")
    assert np.isfinite(y).all() and d.max() <= 2e-6


@pytest.mark.parametrize("shape", [(1, 60, 80), (2, 12, 16), (1, 36, 44), (1, 9, 12)])
def test_pyramid53_kernel_on_the_host(emu_bins, shape):
    """block5.3 (1x1, 128 -> 64, BN folded, ReLU) fused into the pyramid sum (pyramid53_kernel: the default path's form; modules/model.py:78,146-148): x3 + up(x4) +
    up(relu(W y5 + b)) against ATen in float64 -- VGA's maps, small ones, a 1/32 map with more positions than threads' first pass covers only partly (36 x 44 -> 9 x 11),
    odd 1/16 and 1/32 sizes; every output channel group of every image."""
    B, H3, W3 = shape
    H4, W4, H5, W5 = (H3 + 1) // 2, (W3 + 1) // 2, (H3 + 3) // 4, (W3 + 3) // 4
    g = torch.Generator().manual_seed(H3 * W3 + 1)
    x3, x4 = torch.randn(B, 64, H3, W3, generator=g), torch.randn(B, 64, H4, W4, generator=g)
    y5 = torch.randn(B, 128, H5, W5, generator=g)
    w = torch.randn(64, 128, generator=g) * 0.1
    bias = torch.randn(64, generator=g) * 0.3
    out = subprocess.run([emu_bins["pyramid53_emu"]], input=_blob([B, H3, W3, H4, W4, H5, W5, 1], [x3, x4, y5, w.t().contiguous(), bias]), capture_output=True, check=True, timeout=400).stdout
    F = torch.nn.functional
    x5 = torch.relu(torch.einsum("ok,bkhw->bohw", w.double(), y5.double()) + bias.double()[None, :, None, None])
    ref = x3.double() + F.interpolate(x4.double(), (H3, W3), mode="bilinear") + F.interpolate(x5, (H3, W3), mode="bilinear")
    y = np.frombuffer(out, np.float32).reshape(B, 64, H3, W3)
    d = np.abs(y - ref.numpy())
    print(f"pyramid53 {shape}: max |err| {d.max():.3g}, max |x5| {float(x5.abs().max()):.3g}")
    assert np.isfinite(y).all() and d.max() <= 4e-6
    # exact halves / quarters run the row-window form of the interpolation: bit for bit what the table-driven taps give (same operands, same expression)
    out_g = subprocess.run([emu_bins["pyramid53_emu"]], input=_blob([B, H3, W3, H4, W4, H5, W5, 1 | 2], [x3, x4, y5, w.t().contiguous(), bias]), capture_output=True, check=True, timeout=400).stdout
    assert np.array_equal(np.frombuffer(out_g, np.float32).view(np.uint32), np.frombuffer(out, np.float32).view(np.uint32)) or \
        np.array_equal(np.frombuffer(out_g, np.float32), np.frombuffer(out, np.float32))       # (a zero's sign may differ where a zero-weighted tap comes from the neighbouring column)


@pytest.mark.parametrize("shape", [(2, 3, 96, 128), (1, 1, 64, 96), (2, 2, 32, 40), (1, 3, 480, 640)])
def test_gray_stats_kernels_on_the_host(emu_bins, shape):
    """x.mean(1) and InstanceNorm2d(1) (modules/model.py:135-136, 35) as the raw gray image + a per-image {alpha, beta}: three, one and "any" channels (the three instantiations),
    chunks that end inside a 256-thread row, VGA -- against float64"""
    B, C, H, W = shape
    x = torch.rand(B, C, H, W, generator=torch.Generator().manual_seed(C * H)) * 3 - 1
    out = subprocess.run([emu_bins["gray_emu"]], input=_blob([B, C, H, W], [x]), capture_output=True, check=True, timeout=240).stdout
    gray = np.frombuffer(out[:4 * B * H * W], np.float32).reshape(B, H, W)
    coef = np.frombuffer(out[4 * B * H * W:], np.float32).reshape(B, 2)
    gd = x.double().mean(1)
    alpha = 1.0 / torch.sqrt(gd.var((1, 2), unbiased=False) + 1e-5)
    ref = torch.stack([alpha, -gd.mean((1, 2)) * alpha], 1).numpy()
    e_g, e_c = float(np.abs(gray - gd.numpy()).max()), float(np.abs(coef / ref - 1).max())
    print(f"gray_stats {shape}: gray max |err| {e_g:.3g}, coef max rel err {e_c:.3g}")
    assert np.isfinite(gray).all() and e_g <= 3e-7 and e_c <= 1e-6      # (gray: an fp32 sum of C values and one division)


@pytest.mark.parametrize("shape,scale", [((1, 3, 100, 164), 1.3), ((1, 3, 110, 164), 0.6), ((1, 1, 68, 100), 1.0), ((2, 3, 88, 128), 0.77)])
def test_fused_two_stage_resize_kernels_on_the_host(emu_bins, shape, scale):
    """extract_dualscale's F.interpolate(x, scale_factor=s) + preprocess_tensor's resize to multiples of 32 + x.mean(1) + the InstanceNorm statistics (modules/xfeat.py:379-381,
    234-238; modules/model.py:135-136) in ONE kernel: the gather form and the staged form (input region of the NEXT tile in flight, tap tables in LDS, several tiles per
    workgroup, ragged last tiles) must agree BIT FOR BIT, and both with torch's two interpolations to fp32 rounding"""
    B, C, Hin, Win = shape
    x = torch.rand(B, C, Hin, Win, generator=torch.Generator().manual_seed(Hin + C)) * 3 - 1
    Hm, Wm = int(np.floor(Hin * scale)), int(np.floor(Win * scale))
    Ho, Wo = max(32, Hm // 32 * 32), max(32, Wm // 32 * 32)
    s1 = np.float32(1.0 / scale)                                  # (F.interpolate with scale_factor: ATen uses 1 / scale_factor)
    s2h, s2w = np.float32(Hm) / np.float32(Ho), np.float32(Wm) / np.float32(Wo)
    blob = np.array([B, C, Hin, Win, Hm, Wm, Ho, Wo], np.int32).tobytes() + np.array([s1, s1, s2h, s2w], np.float32).tobytes() + x.numpy().astype(np.float32).tobytes()
    out = subprocess.run([emu_bins["resize2_emu"]], input=blob, capture_output=True, check=True, timeout=600).stdout
    n = B * Ho * Wo
    g0 = np.frombuffer(out[:4 * n], np.float32).reshape(B, Ho, Wo)
    g1 = np.frombuffer(out[4 * n:8 * n], np.float32).reshape(B, Ho, Wo)
    c0 = np.frombuffer(out[8 * n:8 * n + 8 * B], np.float32)
    c1 = np.frombuffer(out[8 * n + 8 * B:], np.float32)
    assert np.isfinite(g0).all() and np.array_equal(g0, g1) and np.array_equal(c0, c1)
    mid = torch.nn.functional.interpolate(x, size=(Hm, Wm), scale_factor=None, mode="bilinear", align_corners=False) if scale == 1.0 else \
        torch.nn.functional.interpolate(x, scale_factor=scale, mode="bilinear", align_corners=False)
    ref = torch.nn.functional.interpolate(mid, size=(Ho, Wo), mode="bilinear", align_corners=False).mean(1).numpy()
    err = float(np.abs(g0 - ref).max())
    print(f"resize2 {shape} x {scale}: ({Hm}, {Wm}) -> ({Ho}, {Wo}); the two forms identical; max |gray - torch| {err:.3g}")
    assert tuple(mid.shape[2:]) == (Hm, Wm) and err <= 2e-5      # (ATen's CPU kernel rounds its weights differently; the GPU suite compares with the oracle bit for bit)


@pytest.mark.parametrize("P,N1,N2,n1,n2,nsplit", [(1, 96, 64, 96, 64, 1), (2, 300, 280, 290, 259, 0), (1, 520, 300, 520, 300, 2), (1, 40, 700, 33, 690, 0),
                                                  (1, 96, 64, 96, 64, -1), (2, 300, 280, 290, 259, -1), (1, 40, 700, 33, 690, -1), (1, 1100, 150, 1061, 140, -1), (1, 2200, 40, 2200, 40, -1)])
def test_match_sweep_kernel_on_the_host(emu_bins, P, N1, N2, n1, n2, nsplit):
    """The matcher's filter pass (modules/xfeat.py:327-348 computes D1 @ D2.T; here one fp16-MFMA sweep that keeps only maxima): row / column maxima and the per-block maxima
    R / C of the fp16 product against numpy on the same fp16 numbers -- several column chunks (N2 > 256), row blocks shared out over workgroups (nsplit), partial blocks, valid
    counts below the capacity (the rows / columns beyond them must not leak into any maximum).  nsplit -1: the one-orientation form (mnn_f16_sweep2_kernel), whose C block
    (g, l) holds the maximum over the rows {1024 g + 32 t + l} of a column -- several row groups (N1 > 1024), a last group with residues that have no valid row."""
    g = torch.Generator().manual_seed(N1 + N2)
    a = torch.nn.functional.normalize(torch.randn(P, N1, 64, generator=g), dim=-1) * 256
    b = torch.nn.functional.normalize(torch.randn(P, N2, 64, generator=g), dim=-1) * 256
    a[:, n1:] = 1.0e3; b[:, n2:] = 1.0e3                      # beyond the valid counts: rows that would win every maximum if they were read
    a16, b16 = a.half().float(), b.half().float()
    out = subprocess.run([emu_bins["match_sweep_emu"]], input=_blob([P, N1, N2, n1, n2, nsplit], [a16, b16]), capture_output=True, check=True, timeout=300).stdout
    y = np.frombuffer(out, np.float32)
    one = nsplit < 0
    ncb, nrb = -(-N2 // 32), (-(-N1 // 1024) * 32 if one else -(-N1 // 32))
    rm, cm = y[:P * N1].reshape(P, N1), y[P * N1:P * (N1 + N2)].reshape(P, N2)
    R = y[P * (N1 + N2):P * (N1 + N2) + P * ncb * N1].reshape(P, ncb, N1)
    C = y[P * (N1 + N2) + P * ncb * N1:].reshape(P, nrb, N2)
    S = (a16.double() @ b16.double().transpose(1, 2)).numpy()[:, :n1, :n2]
    tol = 1e-6 * float(np.abs(S).max()) + 1e-3
    assert np.abs(rm[:, :n1] - S.max(2)).max() <= tol and np.abs(cm[:, :n2] - S.max(1)).max() <= tol
    # block maxima travel as fp16, rounded UP (k_match_f16.hip: f16_up): never below the exact maximum, and above it by no more than the rounding's reach
    def block_ok(got, want):
        hi = want + np.abs(want) * 2.0 ** -9 + 4 * 2.0 ** -23 + tol
        return bool((got >= want - tol).all() and (got <= hi).all())
    for cb in range(-(-n2 // 32)):
        assert block_ok(R[:, cb, :n1], S[:, :, cb * 32:(cb + 1) * 32].max(2)), ("R", cb)
    if one:
        for gl in range(nrb):
            g_, l_ = gl >> 5, gl & 31
            rows = np.arange(1024 * g_ + l_, min(1024 * (g_ + 1), n1), 32)
            if 1024 * g_ < n1 and len(rows):       # (a residue without a valid row in a live group: written, from copies of the last valid row, and never read)
                last_blk = min(-(-n1 // 32), 32 * (g_ + 1)) - 1          # the group's last row block: its rows >= n1 are copies of row n1 - 1 (a valid row of ANOTHER block:
                if 32 * last_blk + l_ >= n1:                             # the block maximum can only grow -- the filter stays conservative)
                    rows = np.append(rows, n1 - 1)
                assert block_ok(C[:, gl, :n2], S[:, rows, :].max(1)), ("C", g_, l_)
    else:
        for rb in range(-(-n1 // 32)):
            assert block_ok(C[:, rb, :n2], S[:, rb * 32:(rb + 1) * 32, :].max(1)), ("C", rb)
    print(f"match sweep P {P} {N1} x {N2} (valid {n1} x {n2}): row / column / block maxima within {tol:.3g} of numpy (max |S| {float(np.abs(S).max()):.4g})")


# (main's end-to-end test of the sliced kernels against the reference-made goldens lives in tests/test_default_chain_emulated.py: the kernel bodies are
# headers here, the shipped forms run there as the control of the prepared ones)
