"""Shipped kernels compiled for the HOST and run against float64 references without a GPU: block1_fused_kernel<5> (the dominant kernel), head_f32r_kernel<KP> (the
default heads) and every convolution kernel of the default path (conv_bx_kernel<24, 24>, conv_bxs2_kernel<24>, conv_bx64_kernel, conv_bx64s2_kernel, conv_wino_kernel) --
alone, and chained END TO END against the key-points the unmodified reference wrote into tests/golden/.  The kernel source is SLICED out of the product files (csrc/k_*.hip --
nothing in them is changed for this) and compiled with the host clang against tests/emu/emu.hpp: one host thread per work-item, LDS as a buffer (initialised to NaN patterns),
__syncthreads a barrier, the LDS-DMA a copy, v_mfma_f32_32x32x2_f32 and the lane exchanges emulated.  What it checks: index arithmetic, tile and weight layouts,
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


def _slice_block1():
    t = open(os.path.join(CSRC, "k_conv_direct.hip")).read()
    s = _between(t, "namespace b1 {", "// Split-bf16 MFMA variants of block1_fused_kernel")
    s = _must_sub(s, "__global__ __launch_bounds__(512) void block1_fused_kernel(", "inline void block1_fused_kernel(")
    s = _must_sub(s, "extern __shared__ __attribute__((aligned(16))) float lds[];", "XFH_DYN_LDS(lds);")
    assert "asm" not in s and "<<<" not in s
    return s


def _slice_heads():
    t = open(os.path.join(CSRC, "k_heads.hip")).read()
    head = _between(t, "typedef float f32x16 __attribute__((ext_vector_type(16)));", "// SHIFT (debug, tools/head_soak.py")
    head = _must_sub(head, "typedef __attribute__((address_space(1))) const void* gptr_t;", "typedef const void* gptr_t;")
    head = _must_sub(head, "typedef __attribute__((address_space(3))) void* lptr_t;", "typedef void* lptr_t;")
    k = _between(t, "// The f32-MFMA heads without the activation tile", "// The same heads on the bf16 matrix cores with three-way split operands")
    k = k[:k.rindex("// ----")]                                                          # (the next section's rule)
    k = _must_sub(k, "__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void head_f32r_kernel(HeadArgs a) {", "inline void head_f32r_kernel(HeadArgs a) {")
    k = _must_sub(k, "    code_shift<SHIFT>();\n", "")
    k = _must_sub(k, "extern __shared__ __attribute__((aligned(16))) float smem_r[];", "XFH_DYN_LDS(smem_r);")
    assert "asm" not in head + k and "<<<" not in head + k
    return head + "\n// " + k


def _slice_conv_bx64():
    """conv_bx64_kernel: the inline assembly (LDS-DMA by buffer_load ... lds, waits, idle slots, register keep-alives) becomes emulator calls or nothing"""
    t = open(os.path.join(CSRC, "k_conv_bx64.hip")).read()
    s = _between(t, "typedef int i32x4 __attribute__((ext_vector_type(4)));", "template <int CIN, int FUSE, bool FX>\nstatic int run_bx64(")
    s = _must_sub(s, "typedef __attribute__((address_space(3))) void* lptr_t;", "")
    s = _must_sub(s, "__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))\nvoid conv_bx64_kernel(Bx64Args a) {", "inline void conv_bx64_kernel(Bx64Args a) {")
    s = _must_sub(s, "extern __shared__ __attribute__((aligned(16))) unsigned char smem_b64[];", "XFH_DYN_LDS_BYTES(smem_b64);")
    s = _must_sub(s, "auto lds_addr = [](const unsigned char* p) { return (unsigned)(size_t)(lptr_t)p; };", "auto lds_addr = [&](const unsigned char* p) { return (unsigned)(p - smem_b64); };")
    s = _must_sub(s, 'asm volatile("s_mov_b32 m0, %0\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(dma_voff), "s"(rs_w), "s"(soff) : "memory");',
                  "emu::dma_b128_to_lds(m0v, dma_voff, rs_w, soff);")
    n0 = s.count("asm volatile")
    s = s.replace('asm volatile("s_waitcnt vmcnt(0)" ::: "memory");', ";")
    s = re.sub(r'asm volatile\("s_nop 7[^;]*;', ";", s)                                   # idle slots (with or without tied accumulators)
    s = re.sub(r'asm volatile\("" : "\+v"[^;]*;', ";", s)                                 # register keep-alives of the fused 1x1
    assert n0 >= 8 and s.count("asm volatile") == s.count('asm volatile("" ::: "memory");'), "an inline-assembly statement of conv_bx64_kernel is not covered"
    assert "<<<" not in s
    return s


def _slice_conv_bx64s2():
    """conv_bx64s2_kernel (the stride-2 64 -> 64 | 128 layers: the split of the next chunk hand-placed inside the MFMA rows of the current one): the same substitutions"""
    t = open(os.path.join(CSRC, "k_conv_bx64s2.hip")).read()
    s = _between(t, "struct Bx64S2Args {", "template <int NCO, bool W4>\nstatic int run_bx64s2(")
    s = _must_sub(s, "__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))\nvoid conv_bx64s2_kernel(Bx64S2Args a) {", "inline void conv_bx64s2_kernel(Bx64S2Args a) {")
    s = _must_sub(s, "extern __shared__ __attribute__((aligned(16))) unsigned char smem_s2[];", "XFH_DYN_LDS_BYTES(smem_s2);")
    s = _must_sub(s, "auto lds_addr = [](const unsigned char* p) { return (unsigned)(size_t)(lptr_t)p; };", "auto lds_addr = [&](const unsigned char* p) { return (unsigned)(p - smem_s2); };")
    s = _must_sub(s, 'asm volatile("s_mov_b32 m0, %0\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(dma_voff), "s"(rs_w), "s"(soff) : "memory");',
                  "emu::dma_b128_to_lds(m0v, dma_voff, rs_w, soff);")
    n0 = s.count("asm volatile")
    s = s.replace('asm volatile("s_waitcnt vmcnt(0)" ::: "memory");', ";")
    s = re.sub(r'asm volatile\("s_nop 7[^;]*;', ";", s)                                   # idle slots
    s = re.sub(r'asm volatile\("" :: "v"[^;]*;', ";", s)                                  # S2_KEEP: register keep-alives of the just-read fragments
    assert n0 == 4 and "asm volatile" not in s, "an inline-assembly statement of conv_bx64s2_kernel is not covered"
    assert "<<<" not in s
    return "typedef int i32x4 __attribute__((ext_vector_type(4)));\n" + s


def _slice_conv_wino():
    """conv_wino_kernel (Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32: input planes and transformed weights by LDS-DMA into a two-slot ring, output transform through an LDS
    exchange, optional trailing 1x1 with a K-split reduction across waves): the three DMA forms become emulator calls, the LDS address a byte offset"""
    t = open(os.path.join(CSRC, "k_conv_wino.hip")).read()
    s = _between(t, "struct WinoArgs {", "// ------------------------------------------------------------------------------------------\n// host side")
    s = _must_sub(s, "__global__ __launch_bounds__(256 * (CB / NCBW) * (TBG / NTBW)) __attribute__((amdgpu_waves_per_eu(2, 2)))\nvoid conv_wino_kernel(WinoArgs a) {", "inline void conv_wino_kernel(WinoArgs a) {")
    s = _must_sub(s, "extern __shared__ __attribute__((aligned(16))) float smem[];", "XFH_DYN_LDS(smem);")
    s = _must_sub(s, "auto lds_addr = [](const float* p) { return (unsigned)(size_t)(lptr_t)p; };", "auto lds_addr = [&](const float* p) { return (unsigned)((p - smem) * 4); };")
    n0 = s.count("asm volatile")
    s = _must_sub(s, 'asm volatile("s_mov_b32 m0, %0\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(uvoff), "s"(rs_u), "s"(soff) : "memory");', "emu::dma_b128_to_lds(m0v, uvoff, rs_u, soff);")
    s = _must_sub(s, 'asm volatile("s_mov_b32 m0, %0\\n\\ts_nop 0\\n\\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(m0v), "v"(xvoff[s]), "s"(rin), "s"(soff) : "memory");', "emu::dma_b32_to_lds(m0v, xvoff[s], rin, soff);")
    s = _must_sub(s, 'asm volatile("s_mov_b32 m0, %0\\n\\ts_nop 0\\n\\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(uvoff), "s"(rs_w2), "s"(soff) : "memory");', "emu::dma_b128_to_lds(m0v, uvoff, rs_w2, soff);")
    s = s.replace('asm volatile("s_waitcnt vmcnt(0)" ::: "memory");', ";")
    s = re.sub(r'unsigned (\w+); asm volatile\("s_getreg_b32[^;]*;', r"unsigned \1 = 0;", s)      # (the trace's XCC / hardware ids: a.trace is NULL here)
    assert n0 == 6 and "asm volatile" not in s, "an inline-assembly statement of conv_wino_kernel is not covered"
    assert "<<<" not in s
    s = s.replace("#define XFH_PIN __builtin_amdgcn_sched_barrier(0)", "#undef XFH_PIN\n#define XFH_PIN __builtin_amdgcn_sched_barrier(0)")
    return "typedef float f32x16 __attribute__((ext_vector_type(16)));\ntypedef int i32x4 __attribute__((ext_vector_type(4)));\n#define __builtin_amdgcn_s_memrealtime() 0ll\n" + s


def _slice_pyramid():
    """pyramid_sum_kernel (x3 + up(x4) + up(x5): source planes and per-plane coefficient tables in LDS) with the interpolation helpers it shares with the resize kernels"""
    t = open(os.path.join(CSRC, "k_preproc.hip")).read()
    s = _between(t, "__device__ inline void lin_coef(", "__global__ __launch_bounds__(256) void resize_bilinear_kernel(")
    s += _between(t, "__device__ inline float bilerp_at(", "void launch_pyramid_sum(")
    s = _must_sub(s, "__global__ __launch_bounds__(256) void pyramid_sum_kernel(", "inline void pyramid_sum_kernel(")
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


def _slice_match_sweep():
    """mnn_f16_sweep_kernel with its constants and the 16-value maximum tree: the three static LDS arrays become pointers into the emulator's LDS, the register keep-alive an
    empty statement"""
    t = open(os.path.join(CSRC, "k_match_f16.hip")).read()
    s = _between(t, "typedef float f32x16 __attribute__((ext_vector_type(16)));", "// 16 lanes per descriptor row (float4 each), 16 rows per pass")
    s += _between(t, "// maximum of 16 accumulator values as a v_max3_f32 tree (8 ops)", "// E in the units of the scaled product")
    s += _between(t, "constexpr int FT_TILES = FT_COLS / 32;", "// thresholds, once the maxima are complete")
    s = _must_sub(s, "__global__ __launch_bounds__(512) void mnn_f16_sweep_kernel(", "inline void mnn_f16_sweep_kernel(")
    s = _must_sub(s, "__shared__ __attribute__((aligned(16))) _Float16 Dl[FT_COLS * FT_DS];", "_Float16* Dl = reinterpret_cast<_Float16*>(emu::wg->lds_base());")
    s = _must_sub(s, "__shared__ float colx[8][FT_COLS];", "float (*colx)[FT_COLS] = reinterpret_cast<float (*)[FT_COLS]>(emu::wg->lds_base() + sizeof(_Float16) * FT_COLS * FT_DS);")
    s = _must_sub(s, "__shared__ int next_block;", "int& next_block = *reinterpret_cast<int*>(emu::wg->lds_base() + sizeof(_Float16) * FT_COLS * FT_DS + sizeof(float) * 8 * FT_COLS);")
    s = _must_sub(s, "XFH_KEEP_FRAGS(bfrag[ct & 1]);", ";")
    s = _must_sub(s, "nxt = __builtin_amdgcn_readfirstlane(t);", "nxt = emu_bcast0(t);")      # (only lane 0 holds t: a real broadcast, emu.hpp's readfirstlane is the identity)
    assert "asm volatile" not in s and "<<<" not in s and "__shared__" not in s
    return s


def _slice_conv_bx24():
    """conv_bx_kernel<24, 24> (block2.0 / block2.1) and conv_bxs2_kernel<24> (block3.0): weights in registers, one staged halo tile per output tile"""
    t = open(os.path.join(CSRC, "k_conv_bx.hip")).read()
    s = _between(t, "struct BxArgs {", "template <int CIN, bool FX>\nstatic int run_bxs2(")
    for name, args in (("conv_bx_kernel", "BxArgs"), ("conv_bxs2_kernel", "BxS2Args")):
        s = _must_sub(s, f"__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))\nvoid {name}({args} a) {{", f"inline void {name}({args} a) {{")
    assert s.count("extern __shared__ __attribute__((aligned(16))) unsigned char smem_bx[];") == 2
    s = s.replace("extern __shared__ __attribute__((aligned(16))) unsigned char smem_bx[];", "XFH_DYN_LDS_BYTES(smem_bx);")
    n0 = s.count("asm volatile")
    s = _must_sub(s, 'asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(lds_alloc));', "lds_alloc = 0;")      # (which workgroup of a CU this is: a start delay, nothing else)
    s = re.sub(r'asm volatile\("s_nop 7[^;]*;', ";", s)                                   # idle slots tied to the accumulators
    s = re.sub(r'asm volatile\("" : "\+v"[^;]*;', ";", s)                                 # "the wait for the weight loads belongs here": register pins
    assert n0 == 5 and "asm volatile" not in s, "an inline-assembly statement of k_conv_bx.hip is not covered"
    assert "<<<" not in s
    return s


def _slice_weight_split():
    t = open(os.path.join(CSRC, "api.hip")).read()
    return _between(t, "static uint16_t bf16_rne(float f) {", "constexpr float kFxMaxWeight")


def _slice_bx_split():
    t = open(os.path.join(CSRC, "bx_split.hpp")).read()
    s = _between(t, "typedef float f32x16 __attribute__((ext_vector_type(16)));", "}  // namespace xfh")
    return s


@pytest.fixture(scope="module")
def emu_bins():
    if not os.path.exists(CLANG):
        pytest.skip("no host clang")
    td = tempfile.mkdtemp()
    open(os.path.join(td, "block1_slice.hpp"), "w").write(_slice_block1())
    open(os.path.join(td, "heads_slice.hpp"), "w").write(_slice_heads())
    open(os.path.join(td, "conv_bx64_slice.hpp"), "w").write(_slice_conv_bx64())
    open(os.path.join(td, "conv_bx64s2_slice.hpp"), "w").write(_slice_conv_bx64s2())
    open(os.path.join(td, "conv_bx24_slice.hpp"), "w").write(_slice_conv_bx24())
    open(os.path.join(td, "conv_wino_slice.hpp"), "w").write(_slice_conv_wino())
    open(os.path.join(td, "pyramid_slice.hpp"), "w").write(_slice_pyramid())
    open(os.path.join(td, "gray_slice.hpp"), "w").write(_slice_gray())
    open(os.path.join(td, "match_sweep_slice.hpp"), "w").write(_slice_match_sweep())
    open(os.path.join(td, "weight_split_slice.hpp"), "w").write(_slice_weight_split())
    open(os.path.join(td, "bx_split_slice.hpp"), "w").write(_slice_bx_split())
    out = {}
    for name in ("block1_emu", "head_emu", "conv_bx64_emu", "conv_bx64s2_emu", "conv_bx24_emu", "conv_wino_emu", "pyramid_emu", "gray_emu", "match_sweep_emu"):
        out[name] = os.path.join(td, name)
        subprocess.run([CLANG, "-O1", "-w", "-std=c++20", "-pthread", "-I", td, "-I", EMU, os.path.join(EMU, name + ".cpp"), "-o", out[name]], check=True)
    return out


def _blob(hdr, arrs):
    return np.concatenate([np.array(hdr, np.int32).view(np.float32)] + [np.asarray(a, np.float32).reshape(-1) for a in arrs]).tobytes()


def test_block1_kernel_on_the_host_is_the_network(emu_bins):
    F = torch.nn.functional
    for seed, (B, H, W) in enumerate(((1, 64, 64), (2, 96, 160))):        # 2 x 1 full tiles; 3 x 2.5 tiles per image (a partial last column of tiles)
        g = torch.Generator().manual_seed(seed)
        r = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).float()
        w = {"w1": r(4, 1, 3, 3, k=0.5), "b1": r(4, k=0.2), "w2": r(8, 4, 3, 3, k=0.25), "b2": r(8, k=0.2), "w3": r(8, 8, 3, 3, k=0.2), "b3": r(8, k=0.2),
             "w4": r(24, 8, 3, 3, k=0.2), "b4": r(24, k=0.2), "skw": r(24, 1, 1, 1, k=0.5), "skb": r(24, k=0.2)}
        gray = torch.rand(B, 1, H, W, generator=g).float()
        coef = torch.stack([0.5 + torch.rand(B, generator=g) * 3, torch.randn(B, generator=g)], 1).float()      # per-image {alpha, beta} of the instance normalisation
        d = lambda t: t.double()
        x = d(gray) * d(coef[:, 0]).view(-1, 1, 1, 1) + d(coef[:, 1]).view(-1, 1, 1, 1)
        a = F.relu(F.conv2d(x, d(w["w1"]), d(w["b1"]), padding=1))
        a = F.relu(F.conv2d(a, d(w["w2"]), d(w["b2"]), stride=2, padding=1))
        a = F.relu(F.conv2d(a, d(w["w3"]), d(w["b3"]), padding=1))
        a = F.relu(F.conv2d(a, d(w["w4"]), d(w["b4"]), stride=2, padding=1))
        ref = (a + F.conv2d(F.avg_pool2d(x, 4, 4), d(w["skw"]), d(w["skb"]))).numpy()      # modules/model.py:40-48,140
        kc = lambda t: t.permute(1, 2, 3, 0).reshape(-1).contiguous()          # (cout, cin, 3, 3) -> [(ci * 9 + tap) * cout + co]
        pad = lambda t: torch.cat([t.reshape(-1), torch.zeros(32 - t.numel())])
        out = subprocess.run([emu_bins["block1_emu"]], input=_blob([B, H, W, 5], [gray, coef, kc(w["w1"]), w["b1"], kc(w["w2"]), w["b2"], kc(w["w3"]), w["b3"], kc(w["w4"]), pad(w["b4"]),
                                                                               pad(w["skw"]), pad(w["skb"])]), capture_output=True, check=True, timeout=240).stdout
        x1 = np.frombuffer(out, np.float32).reshape(B, 24, H // 4, W // 4)
        dd = np.abs(x1 - ref)
        print(f"block1 ({B},{H},{W}): max |err| {dd.max():.3g}, max |x1| {np.abs(ref).max():.3g}")
        assert np.isfinite(x1).all() and dd.max() <= 2e-5 * max(1.0, float(np.abs(ref).max()))


def test_default_heads_on_the_host(emu_bins):
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 96, 136                              # 2 x 12 x 17 = 408 cells: one full tile and a partial one
    gray = torch.rand(B, H, W, generator=g)
    coef = torch.stack([1.0 + torch.rand(B, generator=g) * 2, torch.randn(B, generator=g) * 0.5], 1)
    ws = [torch.randn(64, 64, generator=g) * 0.18 for _ in range(3)] + [torch.randn(65, 64, generator=g) * 0.3]
    bs = [torch.randn(64, generator=g) * 0.3 for _ in range(3)] + [torch.randn(65, generator=g)]
    out = subprocess.run([emu_bins["head_emu"]], input=_blob([1, B, H, W], [gray, coef] + ws + bs), capture_output=True, check=True, timeout=240).stdout
    ncell = B * (H // 8) * (W // 8)
    heat = np.frombuffer(out[:4 * B * H * W], np.float32).reshape(B, H, W)
    logits = np.frombuffer(out[4 * B * H * W:], np.float32).reshape(ncell, 65)
    # 8 x 8 unfold (channel = 8 dy + dx) -> 3 x (linear + ReLU) -> linear -> softmax, dustbin dropped, depth-to-space   (modules/model.py:87-92,152; modules/xfeat.py:242-247)
    x = gray.double() * coef[:, 0].double().view(-1, 1, 1) + coef[:, 1].double().view(-1, 1, 1)
    a = x.view(B, H // 8, 8, W // 8, 8).permute(0, 1, 3, 2, 4).reshape(ncell, 64)
    for w, b in zip(ws[:3], bs[:3]):
        a = torch.relu(a @ w.double().T + b.double())
    lg = a @ ws[3].double().T + bs[3].double()
    href = torch.softmax(lg, 1)[:, :64].view(B, H // 8, W // 8, 8, 8).permute(0, 1, 3, 2, 4).reshape(B, H, W)
    e_l, e_h = float(np.abs(logits - lg.numpy()).max()), float(np.abs(heat - href.numpy()).max())
    print(f"key-point head: logits max |err| {e_l:.3g} (max |logit| {float(lg.abs().max()):.3g}), heat max |err| {e_h:.3g}")
    assert np.isfinite(heat).all() and e_l <= 2e-5 * float(lg.abs().max()) and e_h <= 1e-6
    # reliability head + 1 / |feats|   (modules/model.py:79-84; modules/xfeat.py:70)
    n = 300
    feats = torch.randn(n, 64, generator=g) * 2
    ws = [torch.randn(64, 64, generator=g) * 0.18 for _ in range(2)]
    w2 = torch.randn(64, generator=g) * 0.2
    bs = [torch.randn(64, generator=g) * 0.3 for _ in range(2)]
    b2 = torch.randn(1, generator=g)
    out = subprocess.run([emu_bins["head_emu"]], input=_blob([0, n, 0, 0], [feats] + ws + [w2] + bs + [b2]), capture_output=True, check=True, timeout=240).stdout
    rel, inv = np.frombuffer(out[:4 * n], np.float32), np.frombuffer(out[4 * n:8 * n], np.float32)
    a = feats.double()
    for w, b in zip(ws, bs):
        a = torch.relu(a @ w.double().T + b.double())
    ref = torch.sigmoid(a @ w2.double() + b2.double())
    iref = 1.0 / feats.double().norm(dim=1).clamp_min(1e-12)
    e_r, e_i = float(np.abs(rel - ref.numpy()).max()), float(np.abs(inv / iref.numpy() - 1).max())
    print(f"reliability head: max |err| {e_r:.3g}, 1 / |feats| max rel err {e_i:.3g}")
    assert e_r <= 2e-6 and e_i <= 1e-6


@pytest.mark.parametrize("fuse,fx,shape,grid", [(0, 1, (1, 24, 40), 3), (0, 0, (1, 16, 16), 2), (1, 1, (2, 24, 32), 5), (2, 1, (1, 18, 20), 2), (0, 1, (8, 8, 16), 8), (0, 1, (1, 10, 22), 2), (1, 1, (1, 9, 17), 3)])
def test_conv_bx64_kernel_on_the_host(emu_bins, fuse, fx, shape, grid):
    """the 64 -> 64 3x3 convolutions on split-operand MFMAs (fp16 pair / bf16 three-way split), alone and with their trailing 1x1 fused, NCHW or channels-last output:
    full tiles, a half tile, partial strips and rows (24 x 40, 18 x 20), widths that are no multiple of four (22, 17: the masked tail of a loaded pixel quad), and the XCD
    mapping of the work list (B = 8 on a grid of 8)"""
    B, H, W = shape
    g = torch.Generator().manual_seed(10 * fuse + fx)
    x = torch.relu(torch.randn(B, 64, H, W, generator=g)) * 2
    w = torch.randn(64, 64, 3, 3, generator=g) / 24
    b = torch.randn(64, generator=g) * 0.3
    w2 = torch.randn(64, 64, generator=g) / 8
    b2 = torch.randn(64, generator=g) * 0.3
    out = subprocess.run([emu_bins["conv_bx64_emu"]], input=_blob([B, H, W, fuse, fx, 1, 0, grid], [x, w, b] + ([w2, b2] if fuse else [])), capture_output=True, check=True, timeout=240).stdout
    y = np.frombuffer(out[:-4], np.float32)
    status = int(np.frombuffer(out[-4:], np.int32)[0])
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1))
    if fuse:
        ref = torch.nn.functional.conv2d(ref, w2.double().view(64, 64, 1, 1), b2.double())
    y = y.reshape(B, H, W, 64).transpose(0, 3, 1, 2) if fuse == 2 else y.reshape(B, 64, H, W)
    d = np.abs(y - ref.numpy())
    print(f"conv_bx64 fuse {fuse} fx {fx} {shape}: max |err| {d.max():.3g}, max |y| {float(ref.abs().max()):.3g}")
    assert status == 0 and np.isfinite(y).all() and d.max() <= 3e-6 * float(ref.abs().max())


@pytest.mark.parametrize("cout,shape,grid", [(64, (1, 16, 32), 1), (64, (2, 30, 40), 3), (128, (1, 30, 40), 2), (64, (1, 9, 11), 1), (128, (1, 18, 22), 2)])
def test_conv_bx64s2_kernel_on_the_host(emu_bins, cout, shape, grid):
    """the stride-2 64 -> 64 | 128 convolutions (block4.0 / block5.0) on bf16 MFMAs with three-way split operands -- the kernel whose barrier waits carried the round-3
    store-ordering bug (a memory-ordering matter the host cannot see: what runs here is its index arithmetic, the parity-separated input buffers, the split
    hand-placed in the MFMA rows, the cyclic weight stream across units): one full unit, partial rows and strips with several units per workgroup, two cout halves,
    odd sizes with W % 4 != 0 (the masked tail of a loaded pixel quad)"""
    B, H, W = shape
    g = torch.Generator().manual_seed(cout + H)
    x = torch.randn(B, 64, H, W, generator=g) * 2
    w = torch.randn(cout, 64, 3, 3, generator=g) / 24
    b = torch.randn(cout, generator=g) * 0.3
    out = subprocess.run([emu_bins["conv_bx64s2_emu"]], input=_blob([B, H, W, cout, 1, grid], [x, w, b]), capture_output=True, check=True, timeout=400).stdout
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1))
    y = np.frombuffer(out, np.float32).reshape(tuple(ref.shape))
    d = np.abs(y - ref.numpy())
    print(f"conv_bx64s2 cout {cout} {shape}: max |err| {d.max():.3g}, max |y| {float(ref.abs().max()):.3g}")
    assert np.isfinite(y).all() and d.max() <= 4e-6 * float(ref.abs().max())


@pytest.mark.parametrize("stride,fx,shape,grid", [(1, 1, (1, 16, 64), 2), (1, 0, (1, 8, 32), 1), (1, 1, (2, 21, 45), 3), (2, 1, (1, 16, 64), 2), (2, 0, (1, 8, 32), 1), (2, 1, (2, 21, 45), 3), (1, 1, (8, 8, 32), 8)])
def test_conv_bx24_kernels_on_the_host(emu_bins, stride, fx, shape, grid):
    """the 24-channel layers on split-operand MFMAs with their weights in registers: conv_bx_kernel<24, 24> (block2.0 / block2.1) and conv_bxs2_kernel<24> (block3.0, stride 2,
    64 couts), in the shipped fp16-pair arithmetic and the bf16 three-way split: full tiles, partial tiles with odd sizes (21 x 45), several tiles per workgroup, the XCD mapping
    of the work list (B = 8 on a grid of 8)"""
    B, H, W = shape
    cout = 64 if stride == 2 else 24
    g = torch.Generator().manual_seed(7 * stride + fx + H)
    x = torch.relu(torch.randn(B, 24, H, W, generator=g)) * 2
    w = torch.randn(cout, 24, 3, 3, generator=g) / 15
    b = torch.randn(cout, generator=g) * 0.3
    out = subprocess.run([emu_bins["conv_bx24_emu"]], input=_blob([B, H, W, stride, fx, 1, grid], [x, w, b]), capture_output=True, check=True, timeout=400).stdout
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=1))
    y = np.frombuffer(out[:-4], np.float32).reshape(tuple(ref.shape))
    d = np.abs(y - ref.numpy())
    print(f"conv_bx24 stride {stride} fx {fx} {shape}: max |err| {d.max():.3g}, max |y| {float(ref.abs().max()):.3g}")
    assert int(np.frombuffer(out[-4:], np.int32)[0]) == 0 and np.isfinite(y).all() and d.max() <= 3e-6 * float(ref.abs().max())


@pytest.mark.parametrize("cin,fuse,shape,tall", [(64, 0, (1, 8, 16), 0), (64, 1, (2, 9, 13), 1), (64, 2, (1, 16, 8), 1), (128, 0, (1, 6, 10), 0), (128, 1, (2, 15, 20), 0), (128, 0, (1, 15, 20), 1)])
def test_conv_wino_kernel_on_the_host(emu_bins, cin, fuse, shape, tall):
    """Winograd F(2x2,3x3) on the f32 matrix cores (block5.1, block5.2 + 5.3 on the default path; every >= 64-channel 3x3/s1 layer under option wino): 64 and 128 channels, alone and
    with the trailing 1x1 fused (NCHW / channels-last output, the K-split reduction across two or four waves), both tile-region shapes, full and partial regions, odd sizes"""
    B, H, W = shape
    g = torch.Generator().manual_seed(cin + fuse)
    x = torch.relu(torch.randn(B, cin, H, W, generator=g)) * 2
    w = torch.randn(cin, cin, 3, 3, generator=g) / (3 * cin ** 0.5)
    b = torch.randn(cin, generator=g) * 0.3
    w2 = torch.randn(64, cin, generator=g) / cin ** 0.5
    b2 = torch.randn(64, generator=g) * 0.3
    out = subprocess.run([emu_bins["conv_wino_emu"]], input=_blob([B, H, W, cin, fuse, 1, 0, tall], [x, w, b] + ([w2, b2] if fuse else [])), capture_output=True, check=True, timeout=600).stdout
    ref = torch.relu(torch.nn.functional.conv2d(x.double(), w.double(), b.double(), padding=1))
    if fuse:
        ref = torch.nn.functional.conv2d(ref, w2.double().view(64, cin, 1, 1), b2.double())
    y = np.frombuffer(out, np.float32)
    y = y.reshape(B, H, W, 64).transpose(0, 3, 1, 2) if fuse == 2 else y.reshape(tuple(ref.shape))
    d = np.abs(y - ref.numpy())
    print(f"conv_wino cin {cin} fuse {fuse} {shape} tall {tall}: max |err| {d.max():.3g}, max |y| {float(ref.abs().max()):.3g}")
    assert np.isfinite(y).all() and d.max() <= 1e-6 * float(ref.abs().max())      # (Winograd's transforms cost a few ulps: DESIGN 3.2)


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


@pytest.mark.parametrize("P,N1,N2,n1,n2,nsplit", [(1, 96, 64, 96, 64, 1), (2, 300, 280, 290, 259, 0), (1, 520, 300, 520, 300, 2), (1, 40, 700, 33, 690, 0)])
def test_match_sweep_kernel_on_the_host(emu_bins, P, N1, N2, n1, n2, nsplit):
    """The matcher's filter pass (modules/xfeat.py:327-348 computes D1 @ D2.T; here one fp16-MFMA sweep that keeps only maxima): row / column maxima and the per-block maxima
    R / C of the fp16 product against numpy on the same fp16 numbers -- several column chunks (N2 > 256), row blocks shared out over workgroups (nsplit), partial blocks, valid
    counts below the capacity (the rows / columns beyond them must not leak into any maximum)."""
    g = torch.Generator().manual_seed(N1 + N2)
    a = torch.nn.functional.normalize(torch.randn(P, N1, 64, generator=g), dim=-1) * 256
    b = torch.nn.functional.normalize(torch.randn(P, N2, 64, generator=g), dim=-1) * 256
    a[:, n1:] = 1.0e3; b[:, n2:] = 1.0e3                      # beyond the valid counts: rows that would win every maximum if they were read
    a16, b16 = a.half().float(), b.half().float()
    out = subprocess.run([emu_bins["match_sweep_emu"]], input=_blob([P, N1, N2, n1, n2, nsplit], [a16, b16]), capture_output=True, check=True, timeout=300).stdout
    y = np.frombuffer(out, np.float32)
    ncb, nrb = -(-N2 // 32), -(-N1 // 32)
    rm, cm = y[:P * N1].reshape(P, N1), y[P * N1:P * (N1 + N2)].reshape(P, N2)
    R = y[P * (N1 + N2):P * (N1 + N2) + P * ncb * N1].reshape(P, ncb, N1)
    C = y[P * (N1 + N2) + P * ncb * N1:].reshape(P, nrb, N2)
    S = (a16.double() @ b16.double().transpose(1, 2)).numpy()[:, :n1, :n2]
    tol = 1e-6 * float(np.abs(S).max()) + 1e-3
    assert np.abs(rm[:, :n1] - S.max(2)).max() <= tol and np.abs(cm[:, :n2] - S.max(1)).max() <= tol
    for cb in range(-(-n2 // 32)):
        assert np.abs(R[:, cb, :n1] - S[:, :, cb * 32:(cb + 1) * 32].max(2)).max() <= tol, ("R", cb)
    for rb in range(-(-n1 // 32)):
        assert np.abs(C[:, rb, :n2] - S[:, rb * 32:(rb + 1) * 32, :].max(1)).max() <= tol, ("C", rb)
    print(f"match sweep P {P} {N1} x {N2} (valid {n1} x {n2}): row / column / block maxima within {tol:.3g} of numpy (max |S| {float(np.abs(S).max()):.4g})")


@pytest.mark.parametrize("which,convs", [("g1_small", False), ("g2_vga_pair", False), ("g1_small", True), ("g1_small", "single"),
                                         pytest.param("g2_vga_pair", True, marks=pytest.mark.skipif(not os.environ.get("XFH_EMU_VGA"), reason="ten minutes of emulation: XFH_EMU_VGA=1 (log: profiles/r04_emulated_end_to_end_vga.txt)"))])
def test_shipped_kernels_on_the_host_keep_the_references_key_points(emu_bins, which, convs):
    """The three sliced kernels END TO END against the reference-made goldens, without a GPU: block1_fused_kernel<5> and both default heads run in the host emulation on the
    golden fixtures' images and weights (BatchNorm folded here the way xfh_create folds it), everything between and after them (block2 .. feats; NMS, scores, top-k,
    descriptors) is the oracle's fp32 restatement, and the key-point lists are compared -- by the GPU suite's own comparator -- with what the UNMODIFIED reference wrote into
    tests/golden/ (g1_small: 2 x 256 key-points; g2_vga_pair: 2 x 4096 at VGA).  The same key-point SET as the reference; rank moves only among scores a few ulps apart.
    With `convs` the split-operand convolution kernels join in the routing of the bench batch (fp16-pair arithmetic): block2.0 / 2.1 and block3.0 (conv_bx_kernel<24, 24>,
    conv_bxs2_kernel<24>), block3.1 + 3.2, block4.1, block4.2, block_fusion.0, block_fusion.1 + .2 (conv_bx64_kernel, all three fused forms), block4.0 and block5.0
    (conv_bx64s2_kernel), block5.1 and block5.2 + 5.3 (conv_wino_kernel) -- ALL 17 convolution layers of the path behind block1, plus block1 and the heads, as sliced product
    source, pyramid_sum_kernel between them and gray_stats_kernel + gray_coef_kernel in front: the whole network from the image to feats / heat / reliability; what stays with the oracle is the detection (NMS, scores, top-k, descriptors).
    convs = "single": the routing of a single frame or a small batch (the reference's headline use, realtime_demo.py) -- every 64 -> 64 3x3 layer on conv_wino_kernel (alone, with the
    trailing 1x1 fused, channels-last) instead of conv_bx64_kernel, everything else as above."""
    import sys
    import torch.nn.functional as F
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fixtures
    import parity
    from oracle import xfeat_oracle as O

    def fold(sd, name):      # conv (no bias) + eval BatchNorm (affine=False, eps 1e-5) -> weight, bias   (modules/model.py:16-22)
        s_ = 1.0 / torch.sqrt(sd[f"{name}.layer.1.running_var"].double() + 1e-5)
        return (sd[f"{name}.layer.0.weight"].double() * s_.view(-1, 1, 1, 1)).float(), (-sd[f"{name}.layer.1.running_mean"].double() * s_).float()

    def run_block1(sd, gray, coef):
        B, H, W = gray.shape
        (w1, b1), (w2, b2), (w3, b3), (w4, b4) = (fold(sd, f"block1.{i}") for i in range(4))
        kc = lambda t: t.permute(1, 2, 3, 0).reshape(-1).contiguous()
        pad = lambda t: torch.cat([t.reshape(-1), torch.zeros(32 - t.numel())])
        out = subprocess.run([emu_bins["block1_emu"]], input=_blob([B, H, W, 5], [gray, coef, kc(w1), b1, kc(w2), b2, kc(w3), b3, kc(w4), pad(b4), pad(sd["skip1.1.weight"].float()),
                                                                                   pad(sd["skip1.1.bias"].float())]), capture_output=True, check=True, timeout=600).stdout
        return torch.from_numpy(np.frombuffer(out, np.float32).reshape(B, 24, H // 4, W // 4).copy())

    def run_kp_head(sd, gray, coef):
        B, H, W = gray.shape
        ws, bs = zip(*[(w.view(64, 64), b) for w, b in (fold(sd, f"keypoint_head.{i}") for i in range(3))])
        ws, bs = list(ws) + [sd["keypoint_head.3.weight"].view(65, 64).float()], list(bs) + [sd["keypoint_head.3.bias"].float()]
        out = subprocess.run([emu_bins["head_emu"]], input=_blob([1, B, H, W], [gray, coef] + ws + bs), capture_output=True, check=True, timeout=600).stdout
        return torch.from_numpy(np.frombuffer(out[:4 * B * H * W], np.float32).reshape(B, 1, H, W).copy())

    def run_rel_head(sd, feats):
        B, _, h, w = feats.shape
        cl = feats.permute(0, 2, 3, 1).reshape(-1, 64).contiguous()
        ws, bs = zip(*[(w_.view(64, 64), b_) for w_, b_ in (fold(sd, f"heatmap_head.{i}") for i in range(2))])
        out = subprocess.run([emu_bins["head_emu"]], input=_blob([0, len(cl), 0, 0], [cl] + list(ws) + [sd["heatmap_head.2.weight"].view(64).float()] + list(bs) +
                                                                 [sd["heatmap_head.2.bias"].float()]), capture_output=True, check=True, timeout=600).stdout
        return torch.from_numpy(np.frombuffer(out[:4 * len(cl)], np.float32).reshape(B, 1, h, w).copy())

    def conv_emu(kind, hdr, x, tensors, shape, status=True, cl=False):
        out = subprocess.run([emu_bins[kind]], input=_blob(hdr, [x] + tensors), capture_output=True, check=True, timeout=3000).stdout
        if status:
            assert int(np.frombuffer(out[-4:], np.int32)[0]) == 0, (kind, hdr)
            out = out[:-4]
        y = torch.from_numpy(np.frombuffer(out, np.float32).copy())
        return y.view(shape[0], shape[2], shape[3], shape[1]).permute(0, 3, 1, 2).contiguous() if cl else y.view(shape)

    def wino_tall(H_, W_):      # launch_conv_wino's choice of the tile region (8 x 4 instead of 4 x 8 where that needs fewer workgroups)
        return int(-(-((H_ + 1) // 2) // 8) * -(-((W_ + 1) // 2) // 4) < -(-((H_ + 1) // 2) // 4) * -(-((W_ + 1) // 2) // 8))

    def middle_emulated(sd, x1):      # the same layers on the sliced kernels; grids chosen so that workgroups walk several tiles / units
        B, _, H4, W4 = x1.shape
        H8, W8, H16, W16, H32, W32 = H4 // 2, W4 // 2, H4 // 4, W4 // 4, H4 // 8, W4 // 8
        if convs == "single":      # the routing of a single frame or a small batch (api.hip: conv_mfma_checked with maps below the "large map" mark): every 64 -> 64 layer on Winograd
            a = conv_emu("conv_bx24_emu", [B, H4, W4, 1, 1, 1, 5], x1, list(fold(sd, "block2.0")), (B, 24, H4, W4))
            a = conv_emu("conv_bx24_emu", [B, H4, W4, 1, 1, 1, 5], a, list(fold(sd, "block2.1")), (B, 24, H4, W4))
            x3 = conv_emu("conv_bx24_emu", [B, H4, W4, 2, 1, 1, 5], a, list(fold(sd, "block3.0")), (B, 64, H8, W8))
            w2, b2 = fold(sd, "block3.2")
            x3 = conv_emu("conv_wino_emu", [B, H8, W8, 64, 1, 1, 1, wino_tall(H8, W8)], x3, list(fold(sd, "block3.1")) + [w2.view(64, 64), b2], (B, 64, H8, W8), status=False)
            x4 = conv_emu("conv_bx64s2_emu", [B, H8, W8, 64, 1, 3], x3, list(fold(sd, "block4.0")), (B, 64, H16, W16), status=False)
            x4 = conv_emu("conv_wino_emu", [B, H16, W16, 64, 0, 1, 0, wino_tall(H16, W16)], x4, list(fold(sd, "block4.1")), (B, 64, H16, W16), status=False)
            x4 = conv_emu("conv_wino_emu", [B, H16, W16, 64, 0, 1, 0, wino_tall(H16, W16)], x4, list(fold(sd, "block4.2")), (B, 64, H16, W16), status=False)
            x5 = conv_emu("conv_bx64s2_emu", [B, H16, W16, 128, 1, 2], x4, list(fold(sd, "block5.0")), (B, 128, H32, W32), status=False)
            x5 = conv_emu("conv_wino_emu", [B, H32, W32, 128, 0, 1, 0, wino_tall(H32, W32)], x5, list(fold(sd, "block5.1")), (B, 128, H32, W32), status=False)
            w2, b2 = fold(sd, "block5.3")
            x5 = conv_emu("conv_wino_emu", [B, H32, W32, 128, 1, 1, 1, wino_tall(H32, W32)], x5, list(fold(sd, "block5.2")) + [w2.view(64, 128), b2], (B, 64, H32, W32), status=False)
            f = conv_emu("pyramid_emu", [B * 64, H8, W8, H16, W16, H32, W32, 1], x3, [x4, x5], (B, 64, H8, W8), status=False)
            f = conv_emu("conv_wino_emu", [B, H8, W8, 64, 0, 1, 0, wino_tall(H8, W8)], f, list(fold(sd, "block_fusion.0")), (B, 64, H8, W8), status=False)
            return conv_emu("conv_wino_emu", [B, H8, W8, 64, 2, 1, 0, wino_tall(H8, W8)], f, list(fold(sd, "block_fusion.1")) + [sd["block_fusion.2.weight"].view(64, 64).float(), sd["block_fusion.2.bias"].float()],
                            (B, 64, H8, W8), status=False, cl=True)
        a = conv_emu("conv_bx24_emu", [B, H4, W4, 1, 1, 1, 5], x1, list(fold(sd, "block2.0")), (B, 24, H4, W4))
        a = conv_emu("conv_bx24_emu", [B, H4, W4, 1, 1, 1, 5], a, list(fold(sd, "block2.1")), (B, 24, H4, W4))
        x3 = conv_emu("conv_bx24_emu", [B, H4, W4, 2, 1, 1, 5], a, list(fold(sd, "block3.0")), (B, 64, H8, W8))
        w2, b2 = fold(sd, "block3.2")
        x3 = conv_emu("conv_bx64_emu", [B, H8, W8, 1, 1, 1, 1, 5], x3, list(fold(sd, "block3.1")) + [w2.view(64, 64), b2], (B, 64, H8, W8))      # + the 1x1 BasicLayer (ReLU) fused
        x4 = conv_emu("conv_bx64s2_emu", [B, H8, W8, 64, 1, 3], x3, list(fold(sd, "block4.0")), (B, 64, H16, W16), status=False)
        x4 = conv_emu("conv_bx64_emu", [B, H16, W16, 0, 1, 1, 0, 3], x4, list(fold(sd, "block4.1")), (B, 64, H16, W16))
        x4 = conv_emu("conv_bx64_emu", [B, H16, W16, 0, 1, 1, 0, 3], x4, list(fold(sd, "block4.2")), (B, 64, H16, W16))
        x5 = conv_emu("conv_bx64s2_emu", [B, H16, W16, 128, 1, 2], x4, list(fold(sd, "block5.0")), (B, 128, H32, W32), status=False)
        tall = wino_tall(H32, W32)
        x5 = conv_emu("conv_wino_emu", [B, H32, W32, 128, 0, 1, 0, tall], x5, list(fold(sd, "block5.1")), (B, 128, H32, W32), status=False)
        w2, b2 = fold(sd, "block5.3")
        x5 = conv_emu("conv_wino_emu", [B, H32, W32, 128, 1, 1, 1, tall], x5, list(fold(sd, "block5.2")) + [w2.view(64, 128), b2], (B, 64, H32, W32), status=False)      # + block5.3 (1x1 BasicLayer) fused
        f = conv_emu("pyramid_emu", [B * 64, H8, W8, H16, W16, H32, W32, 1], x3, [x4, x5], (B, 64, H8, W8), status=False)      # pyramid_sum_kernel (modules/model.py:146-148)
        f = conv_emu("conv_bx64_emu", [B, H8, W8, 0, 1, 1, 0, 5], f, list(fold(sd, "block_fusion.0")), (B, 64, H8, W8))
        return conv_emu("conv_bx64_emu", [B, H8, W8, 2, 1, 1, 0, 5], f, list(fold(sd, "block_fusion.1")) + [sd["block_fusion.2.weight"].view(64, 64).float(), sd["block_fusion.2.bias"].float()],
                        (B, 64, H8, W8), cl=True)      # + the plain 1x1 fused, channels-last output (= feats as the samplers read them)

    def middle(sd, x1):      # block2 .. block_fusion.2 (modules/model.py:141-150), the oracle's own layers
        a = O._basic(sd, "block2.1", O._basic(sd, "block2.0", x1))
        x3 = O._basic(sd, "block3.2", O._basic(sd, "block3.1", O._basic(sd, "block3.0", a, 2)), 1, 1)
        x4 = O._basic(sd, "block4.2", O._basic(sd, "block4.1", O._basic(sd, "block4.0", x3, 2)))
        x5 = O._basic(sd, "block5.3", O._basic(sd, "block5.2", O._basic(sd, "block5.1", O._basic(sd, "block5.0", x4, 2))), 1, 1)
        hw = tuple(x3.shape[-2:])
        f = x3 + F.interpolate(x4, hw, mode="bilinear") + F.interpolate(x5, hw, mode="bilinear")
        return O._plain(sd, "block_fusion.2", O._basic(sd, "block_fusion.1", O._basic(sd, "block_fusion.0", f)))

    def detect(feats, heat, rel, top_k, H, W):      # modules/xfeat.py:70-96 on given maps, with the oracle's pieces
        B = feats.shape[0]
        fn = F.normalize(feats, dim=1)
        mk = O.pad_keypoints(O.nms(heat, 0.05, 5))
        scores = torch.stack([O.sample_nearest(heat[b], mk[b], H, W)[:, 0] * O.sample_bilinear(rel[b], mk[b], H, W)[:, 0] for b in range(B)])
        scores[torch.all(mk == 0, dim=-1)] = -1
        order = torch.argsort(-scores)
        mk = torch.gather(mk, 1, order[..., None].expand(-1, -1, 2))[:, :top_k]
        scores = torch.gather(scores, 1, order)[:, :top_k]
        desc = F.normalize(torch.stack([O.sample_bicubic(fn[b], mk[b], H, W) for b in range(B)]), dim=-1)
        return [{"keypoints": mk[b][scores[b] > 0].float(), "scores": scores[b][scores[b] > 0], "descriptors": desc[b][scores[b] > 0]} for b in range(B)]

    sd = fixtures.synthetic_state_dict(0)
    with torch.inference_mode():
        for which in (which,):
            g = np.load(os.path.join(ROOT, "tests", "golden", which + ".npz"))
            if which == "g1_small":
                x, top_k = fixtures.texture_images(2, 96, 128, seed=11), 256
                gold = [{k: g[f"{k}{b}"] for k in ("keypoints", "scores", "descriptors")} for b in range(2)]
            else:
                x, top_k = torch.cat(fixtures.shifted_pair(1, 480, 640, seed=7)), 4096
                gold = [{"keypoints": g[f"kp_{t}"].astype(np.float32), "scores": g[f"sc_{t}"]} for t in ("a", "b")]
            B, Cc, H, W = x.shape
            if convs:      # the front of the path too: gray_stats_kernel + gray_coef_kernel
                o_ = subprocess.run([emu_bins["gray_emu"]], input=_blob([B, Cc, H, W], [x]), capture_output=True, check=True, timeout=600).stdout
                gray = torch.from_numpy(np.frombuffer(o_[:4 * B * H * W], np.float32).reshape(B, H, W).copy())
                coef = torch.from_numpy(np.frombuffer(o_[4 * B * H * W:], np.float32).reshape(B, 2).copy())
            else:
                gray = x.mean(1)
                gd = gray.double()
                alpha = 1.0 / torch.sqrt(gd.var((1, 2), unbiased=False) + 1e-5)               # InstanceNorm2d(1) as x * alpha + beta   (modules/model.py:35,136)
                coef = torch.stack([alpha, -gd.mean((1, 2)) * alpha], 1).float()
            _, _, _, taps = O.backbone(sd, x, keep=True)
            oheat = O.kpts_heatmap(taps["logits"])
            x1 = run_block1(sd, gray, coef)
            feats = middle_emulated(sd, x1) if convs else middle(sd, x1)
            rel, heat = run_rel_head(sd, feats), run_kp_head(sd, gray, coef)
            e = {"x1": float((x1 - taps["x1"]).abs().max()), "feats": float((feats - taps["feats"]).abs().max()),
                 "rel": float((rel - taps["reliability"]).abs().max()), "heat": float((heat - oheat).abs().max())}
            print(which, f"convolutions emulated ({convs})" if convs else "convolutions: oracle", e)
            assert e["x1"] <= 2e-5 and e["feats"] <= 1e-4 and e["rel"] <= 3e-5 and e["heat"] <= 1e-5, e      # the GPU suite's tolerances against the oracle
            for b, out in enumerate(detect(feats, heat, rel, top_k, H, W)):
                gd_, t = dict(gold[b]), dict(out)
                if "descriptors" not in gd_:      # (g2 holds every 8th descriptor row only: the lists are compared)
                    gd_["descriptors"] = np.zeros((len(gd_["keypoints"]), 64), np.float32); t["descriptors"] = torch.zeros(len(t["keypoints"]), 64)
                rep = parity.compare_keypoints(t, gd_, heat=oheat[b, 0])      # raises on anything that is not a tie in the reference's own maps
                print(which, "image", b, rep)
                assert rep["common"] == rep["n_ref"] == top_k and rep["exceptions"] == 0, rep
                assert rep.get("rank_moved", 0) <= 64 and rep.get("rank_moved_maxgap", 0.0) <= 5e-6, rep

