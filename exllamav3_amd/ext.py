"""
Host-side mirror of the reference's native op surface `exllamav3_ext` for the EXL3 quantized-linear hot path
(/root/reference/exllamav3/exllamav3_ext/bindings.cpp:69-226).  Same names, argument order and meaning, same
error behaviour (argument violations raise RuntimeError); every op forwards raw device pointers to the C-ABI
library libexl3_hip.so (include/exl3_hip.h) on the caller's current HIP stream.  PyTorch is used only for device
memory and streams.  There is no CPU or eager fallback: a missing library raises at first use.

INTEGRATION.md shows how this module is installed under the name `exllamav3_ext` so that the reference's
`exllamav3/ext.py:20-30` import seam picks it up unchanged.
"""
from __future__ import annotations
import ctypes
import torch
from . import _lib
from ._lib import check as _check

_vp = ctypes.c_void_p


def _p(t: torch.Tensor | None):
    return None if t is None else t.data_ptr()


def _stream(t: torch.Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


def _cb(mcg: bool, mul1: bool) -> int:
    # reconstruct.cu:128-130: mcg wins over mul1
    return 1 if mcg else (2 if mul1 else 0)


def _req(cond: bool, msg: str):
    if not cond:
        raise RuntimeError(msg)


def _dev(t: torch.Tensor):
    _req(t.is_cuda, "tensor must be on a GPU device")
    if t.device.index != torch.cuda.current_device():
        torch.cuda.set_device(t.device)


def init(device: int | torch.device | None = None):
    """Create the per-device context (split-k workspace).  Must run once before graph capture."""
    if device is None:
        device = torch.cuda.current_device()
    if isinstance(device, torch.device):
        device = device.index if device.index is not None else torch.cuda.current_device()
    torch.cuda.set_device(device)
    _check(_lib.lib().exl3_init(int(device)))


def g_get_num_sms(device: int) -> int:
    n = ctypes.c_int(0)
    _check(_lib.lib().exl3_device_info(int(device), ctypes.byref(n), None, None))
    return n.value


def g_get_cc(device: int) -> int:
    a = ctypes.c_int(0)
    _check(_lib.lib().exl3_device_info(int(device), None, ctypes.byref(a), None))
    return a.value


def set_gemv_variant(v: int):
    """0 = EXACT (reference fp16 weights bit-for-bit in the MFMA), 1 = FAST (default)."""
    _lib.lib().exl3_set_gemv_variant(int(v))


def set_gemv_max_waves(n: int):
    """Cap on waves per workgroup of the gen-2 GEMV (0 = heuristic, up to 16)."""
    _lib.lib().exl3_set_gemv_max_waves(int(n))


def set_gemv_gen4(on: bool):
    """1..4-row launches: generation-4 kernel (default) or generation 2 (A/B runs, bit-identity tests between generation-2 pipelines)."""
    _lib.lib().exl3_set_gemv_gen4(int(bool(on)))


def set_tail_xcd_local(on: bool):
    """Opt in to the XCD-local tail hand-off of exl3_gemv_resid (default off: agent-scope hand-off, placement-independent).  Enabling probes the
    workgroup -> XCD mapping the mode relies on and raises RuntimeError (mode stays off) if the device does not dispatch block i to XCD i % 8."""
    _check(_lib.lib().exl3_set_tail_xcd_local(int(bool(on))))


def exl3_gemv_resid(A, xh, xsum, B, suh, svh, bias, m: int, mcg: bool, mul1: bool, resid, ss_out, force_split: int = 0):
    """o_proj / down_proj whose launch also does glue_resid: resid += linear(x) (fp16, in place), ss_out[m][n/128] = per-block sums of squares.
    Input: raw A (+ suh) or pre-rotated xh."""
    ref = A if A is not None else xh
    _dev(ref)
    k, K = _kK(B)
    n = B.shape[1] * 16
    _req(ref.shape[-1] == k, "exl3_gemv_resid: input width must match B (k)")
    _req(resid.dtype == torch.half and resid.shape[-1] == n and ss_out.dtype == torch.float and ss_out.numel() >= m * (n // 128), "exl3_gemv_resid: bad resid / ss_out")
    _check(_lib.lib().exl3_gemv_resid(_p(A), _p(xh), _p(xsum), _p(B), _p(suh), _p(svh), _p(bias), m, k, n, K, _cb(mcg, mul1), _p(resid), _p(ss_out),
                                      int(force_split), _stream(ref)))


def set_gemm3_min_rows(n: int):
    """Passes with at least n rows use the generation-3 small-m GEMM (exl3_gemm3.kspec.hip); default 9, 0 = never."""
    _lib.lib().exl3_set_gemm3_min_rows(int(n))


def set_gemm3_cpw(n: int):
    """generation 3 (5..16 rows): column blocks per workgroup (1 / 2 / 4 = 4- / 8- / 16-wave workgroups sharing one activation tile); 0 = cost model."""
    _lib.lib().exl3_set_gemm3_cpw(int(n))


def set_attn_wide_waves(n: int):
    """decode attention, matrix-pipe kernel: waves per workgroup (4 / 8 = one / two token-step chains per SIMD); 0 = by the split length."""
    _lib.lib().exl3_set_attn_wide_waves(int(n))


# --------------------------------------------------------------------------------------------------
# format ops
# --------------------------------------------------------------------------------------------------

def pack_trellis(packed: torch.Tensor, unpacked: torch.Tensor, K: int):
    """quant/pack.cu:68-95"""
    _dev(unpacked)
    _req(packed.dim() == 3 and unpacked.dim() == 3, "pack_trellis: tensors must be 3-D")
    _req(packed.shape[0] == unpacked.shape[0] and packed.shape[1] == unpacked.shape[1], "pack_trellis: shape mismatch")
    _req(unpacked.shape[2] == 256, "pack_trellis: unpacked dim 2 must be 256")
    _req(packed.shape[2] == 256 * K // 16, "pack_trellis: packed dim 2 must be 16*K")
    _req(packed.is_contiguous() and unpacked.is_contiguous(), "pack_trellis: tensors must be contiguous")
    _check(_lib.lib().exl3_pack_trellis(_p(packed), _p(unpacked), packed.shape[0], packed.shape[1], K, _stream(unpacked)))


def unpack_trellis(unpacked: torch.Tensor, packed: torch.Tensor, K: int):
    """quant/pack.cu:148-175"""
    _dev(packed)
    _req(packed.dim() == 3 and unpacked.dim() == 3, "unpack_trellis: tensors must be 3-D")
    _req(packed.shape[0] == unpacked.shape[0] and packed.shape[1] == unpacked.shape[1], "unpack_trellis: shape mismatch")
    _req(unpacked.shape[2] == 256, "unpack_trellis: unpacked dim 2 must be 256")
    _req(packed.shape[2] == 256 * K // 16, "unpack_trellis: packed dim 2 must be 16*K")
    _req(packed.is_contiguous() and unpacked.is_contiguous(), "unpack_trellis: tensors must be contiguous")
    _check(_lib.lib().exl3_unpack_trellis(_p(unpacked), _p(packed), packed.shape[0], packed.shape[1], K, _stream(packed)))


def pack_signs(packed: torch.Tensor, unpacked: torch.Tensor):
    """quant/pack.cu:203-226"""
    _dev(unpacked)
    _req(unpacked.dtype == torch.half, "pack_signs: unpacked must be float16")
    _req(packed.numel() * 16 == unpacked.numel(), "pack_signs: size mismatch")
    _check(_lib.lib().exl3_pack_signs(_p(packed), _p(unpacked), unpacked.numel(), _stream(unpacked)))


def decode(idx: torch.Tensor, out: torch.Tensor, mcg: bool, mul1: bool):
    """quant/quantize.cu:118-168: 16-bit states -> codebook values (fp16 or fp32 out)"""
    _dev(idx)
    _req(idx.dtype in (torch.int16, torch.uint16), "decode: idx must be int16")
    _req(out.dtype in (torch.half, torch.float), "decode: out must be float16 or float32")
    _req(idx.numel() == out.numel(), "decode: size mismatch")
    _check(_lib.lib().exl3_decode(_p(idx), _p(out), idx.numel(), int(out.dtype == torch.float), _cb(mcg, mul1), _stream(idx)))


def _check_packed(unpacked, packed, K):
    _req(packed.dim() == 3 and packed.dtype == torch.int16, "packed must be a 3-D int16 tensor")
    _req(packed.shape[2] == 256 * K // 16, "packed dim 2 must be 16*K")
    _req(unpacked.dtype == torch.half, "unpacked must be float16")
    _req(unpacked.dim() == 2 and unpacked.shape[0] == packed.shape[0] * 16, "unpacked dim 0 must be 16 * packed dim 0")
    _req(unpacked.is_contiguous() and packed.is_contiguous(), "tensors must be contiguous")


def reconstruct_slice(unpacked: torch.Tensor, packed: torch.Tensor, K: int, mcg: bool, mul1: bool, n_offset: int):
    """quant/reconstruct.cu:98-144"""
    _dev(unpacked)
    _check_packed(unpacked, packed, K)
    _check(_lib.lib().exl3_reconstruct(_p(unpacked), _p(packed), packed.shape[0], packed.shape[1], K, _cb(mcg, mul1),
                                       n_offset, unpacked.shape[1], _stream(unpacked)))


def reconstruct(unpacked: torch.Tensor, packed: torch.Tensor, K: int, mcg: bool, mul1: bool):
    """quant/reconstruct.cu:375-386"""
    _req(unpacked.dim() == 2 and packed.dim() == 3 and unpacked.shape[1] == packed.shape[1] * 16,
         "reconstruct: unpacked dim 1 must be 16 * packed dim 1")
    reconstruct_slice(unpacked, packed, K, mcg, mul1, 0)


def reconstruct_had_slice(unpacked: torch.Tensor, packed: torch.Tensor, suh: torch.Tensor, svh: torch.Tensor,
                          K: int, mcg: bool, mul1: bool, n_offset: int):
    """quant/reconstruct.cu:324-373 (svh pre-offset by the caller)"""
    _dev(unpacked)
    _check_packed(unpacked, packed, K)
    _req(suh.dtype == torch.half and svh.dtype == torch.half, "suh/svh must be float16")
    _req(unpacked.shape[0] % 128 == 0, "reconstruct_had_slice: K dimension must be divisible by 128")
    _req(suh.numel() >= unpacked.shape[0] and svh.numel() >= unpacked.shape[1], "suh/svh too small")
    _check(_lib.lib().exl3_reconstruct_had(_p(unpacked), _p(packed), _p(suh), _p(svh), packed.shape[0], packed.shape[1],
                                           K, _cb(mcg, mul1), n_offset, unpacked.shape[1], _stream(unpacked)))


def had_r_128(input: torch.Tensor, output: torch.Tensor, pre_scale: torch.Tensor | None, post_scale: torch.Tensor | None,
              scale: float = 1.0):
    """quant/hadamard.cu:88-173"""
    _dev(input)
    _req(input.shape == output.shape, "had_r_128: shape mismatch")
    _req(input.dim() == 2, "had_r_128: input must be 2-D")
    _req(input.shape[1] % 128 == 0, "had_r_128: dim 1 must be divisible by 128")
    _req(input.dtype in (torch.half, torch.float), "unsupported datatype")
    _req(output.dtype == input.dtype, "had_r_128: output dtype mismatch")
    _req(input.is_contiguous() and output.is_contiguous(), "had_r_128: tensors must be contiguous")
    for s in (pre_scale, post_scale):
        _req(s is None or (s.dtype == torch.half and s.numel() >= input.shape[1]), "had_r_128: bad scale tensor")
    _check(_lib.lib().exl3_had_r_128(_p(input), _p(output), _p(pre_scale), _p(post_scale), float(scale),
                                     input.shape[0], input.shape[1], int(input.dtype == torch.float), _stream(input)))


# --------------------------------------------------------------------------------------------------
# quantized GEMM
# --------------------------------------------------------------------------------------------------

def exl3_gemm_num_kernel_shapes() -> int:
    """quant/exl3_kernel_map.cuh:53-60 has 4 tile shapes; this implementation has one geometry."""
    return 1


def exl3_gemm_shape_compat(shape_idx: int, size_m: int, size_k: int, size_n: int, K: int) -> bool:
    return size_k % 128 == 0 and size_n % 128 == 0 and 1 <= K <= 8


def _gemm_checks(A, B, C, suh, svh):
    _req(A.dtype == torch.half, "A must be float16")
    _req(B.dtype == torch.int16 and B.dim() == 3, "B must be a 3-D int16 trellis tensor")
    _req(C.dtype in (torch.half, torch.float), "C must be float16 or float32")
    _req(A.is_contiguous() and B.is_contiguous() and C.is_contiguous(), "A, B and C must be contiguous")
    k, n = B.shape[0] * 16, B.shape[1] * 16
    _req(A.shape[-1] == k, "A dim -1 must match B")
    _req(C.shape[-1] == n, "C dim -1 must match B")
    m = A.numel() // k
    _req(C.numel() // n == m, "A and C row counts differ")
    _req(B.shape[2] % 16 == 0 and 1 <= B.shape[2] // 16 <= 8, "B dim 2 must be 16*K, K in [1, 8]")
    _req(suh is not None and svh is not None, "suh and svh are required")
    _req(suh.dtype == torch.half and svh.dtype == torch.half, "suh/svh must be float16")
    _req(suh.numel() == k and svh.numel() == n, "suh/svh size mismatch")
    return m, k, n, B.shape[2] // 16


def exl3_gemm(A: torch.Tensor, B: torch.Tensor, C: torch.Tensor, suh: torch.Tensor | None, A_had: torch.Tensor | None,
              svh: torch.Tensor | None, force_shape_idx: int = -1, mcg: bool = False, mul1: bool = False,
              force_num_sms: int = 0, bias: torch.Tensor | None = None, force_split: int = 0) -> int:
    """quant/exl3_gemm.cuh:21-33.  A_had (the reference's scratch for the rotated input) is accepted and ignored:
    the input Hadamard is recomputed per workgroup in LDS.  Returns the kernel id (1)."""
    _dev(A)
    m, k, n, K = _gemm_checks(A, B, C, suh, svh)
    if m == 0:
        return 0
    return _check(_lib.lib().exl3_gemm(_p(A), _p(B), _p(C), _p(suh), _p(svh), _p(bias), m, k, n, K, _cb(mcg, mul1),
                                       int(C.dtype == torch.float), int(force_split), _stream(A)))


def exl3_gemv(A, B, C, suh, A_had, svh, mcg: bool, mul1: bool):
    """quant/exl3_gemv.cuh:30-40 (same kernel family here)"""
    exl3_gemm(A, B, C, suh, A_had, svh, -1, mcg, mul1, 0)


def exl3_mgemm(A: torch.Tensor, B: torch.Tensor, C: torch.Tensor, suh: torch.Tensor, A_had, svh: torch.Tensor, indices, weights,
               K: int, force_shape_idx: int, mcg: int, mul1: int, min_index: int, max_index: int, force_num_sms: int = 0,
               num_tokens: int = 1, size_n_list=None, c_ptrs=None) -> int:
    """quant/exl3_gemm.cuh:58-78 (verbatim argument order).  A [bA, m, k] fp16; B / suh / svh: int64 device tensors of device pointers
    (modules/multilinear.py:30-32, block_sparse_mlp.py); C [bC, m, n] fp16/fp32; indices int64 [*, top] or None; weights fp16 or None.
    A_had (scratch) is not needed here.  size_n_list (int32 [num matrices]) + c_ptrs (int64 [num matrices]): per-matrix output widths and output
    addresses (exl3_gemm.cu:433-447) -- C then only carries the dtype and the widest matrix's width."""
    _dev(A)
    if size_n_list is not None or c_ptrs is not None:
        _req(size_n_list is not None and c_ptrs is not None, "exl3_mgemm: size_n_list requires c_ptrs")
        _req(size_n_list.dtype == torch.int32 and c_ptrs.dtype == torch.long and size_n_list.is_contiguous() and c_ptrs.is_contiguous(),
             "exl3_mgemm: size_n_list must be int32, c_ptrs int64")
        _req(num_tokens == 1 and min_index < 0 and weights is None, "exl3_mgemm: per-matrix widths incompatible with multi-token/filtering/weights")
        _req(A.dtype == torch.half and A.dim() == 3 and C.dim() == 3 and A.shape[1] == C.shape[1] and A.is_contiguous(), "exl3_mgemm: bad A / C")
        _req(B.dtype == torch.long and suh.dtype == torch.long and svh.dtype == torch.long and B.dim() == 1, "exl3_mgemm: B, suh, svh must be int64 pointer tensors")
        _req(C.dtype in (torch.half, torch.float), "exl3_mgemm: C must be float16 or float32")
        bA, m, k = A.shape
        bszm = max(bA, int(c_ptrs.shape[0]))                       # exl3_gemm.cu:446-448
        if indices is not None:
            _req(indices.dtype == torch.long and indices.is_contiguous() and indices.dim() == 2, "exl3_mgemm: indices must be int64 [*, slots]")
            bszm = min(bszm, int(indices.shape[1]))                # exl3_gemm.cu:462-464
        _req(bA in (1, bszm) or bA >= bszm, "exl3_mgemm: A must hold 1 or bszm slots")
        _req(size_n_list.numel() >= (bszm if indices is None else 1) and c_ptrs.numel() >= (bszm if indices is None else 1), "exl3_mgemm: width / pointer lists too short")
        _req(k % 128 == 0 and C.shape[2] % 128 == 0, "exl3_mgemm: k and n must be divisible by 128")
        # 16 rows per launch (the reference's kernel walks the rows 16 at a time too: exl3_gemm_kernel.cuh:186-204); later passes write behind the
        # earlier rows of every matrix: output address + 16 * pass * n_i elements, computed on the device (capturable)
        esz = 4 if C.dtype == torch.float else 2
        for m0 in range(0, m, 16):
            mp = min(16, m - m0)
            Ap = A if m <= 16 else A[:, m0:m0 + mp].contiguous()
            cp = c_ptrs if m0 == 0 else c_ptrs + size_n_list.to(torch.long) * (m0 * esz)
            _check(_lib.lib().exl3_mgemm_indexed_nlist(_p(Ap), 1 if bA == 1 else bszm, _p(B), _p(suh), _p(svh), _p(indices), bszm, _p(size_n_list), _p(cp),
                                                       mp, k, int(C.shape[2]), int(K), _cb(bool(mcg), bool(mul1)), int(C.dtype == torch.float), _stream(A)))
        return 90
    _req(A.dtype == torch.half, "exl3_mgemm: A must be float16")
    _req(B.dtype == torch.long and suh.dtype == torch.long and svh.dtype == torch.long, "exl3_mgemm: B, suh, svh must be int64 pointer tensors")
    _req(C.dtype in (torch.half, torch.float), "exl3_mgemm: C must be float16 or float32")
    _req(A.dim() == 3 and C.dim() == 3 and B.dim() == 1 and suh.dim() == 1 and svh.dim() == 1, "exl3_mgemm: bad tensor ranks")
    _req(A.shape[1] == C.shape[1], "exl3_mgemm: A and C must share m")
    _req(B.shape[0] == suh.shape[0] == svh.shape[0], "exl3_mgemm: pointer tables must share a length")
    _req(A.is_contiguous() and C.is_contiguous(), "exl3_mgemm: A and C must be contiguous")
    _req(num_tokens == 1 or min_index < 0, "exl3_mgemm: multi-token reduction (num_tokens > 1) is not compatible with expert-range "
                                           "filtering (min_index >= 0); TP-sharded experts must use num_tokens == 1")
    bA, m, k = A.shape
    bC, _, n = C.shape
    bszm = max(bA, bC)
    _req(bA in (1, bszm) and bC == bszm, "exl3_mgemm: A must hold 1 or bszm slots, C bszm slots")
    _req(k % 128 == 0 and n % 128 == 0, "exl3_mgemm: k and n must be divisible by 128")
    if indices is not None:
        _req(indices.dtype == torch.long and indices.numel() >= bszm and indices.is_contiguous(), "exl3_mgemm: indices must be int64 with >= bszm entries")
    if weights is not None:
        _req(weights.dtype == torch.half and weights.numel() >= bszm and weights.is_contiguous(), "exl3_mgemm: weights must be float16 with >= bszm entries")
    _check(_lib.lib().exl3_mgemm_indexed(_p(A), bA, _p(B), _p(suh), _p(svh), _p(indices), _p(weights), bszm, _p(C), m, k, n, int(K),
                                         _cb(bool(mcg), bool(mul1)), int(C.dtype == torch.float), int(min_index), int(max_index),
                                         int(num_tokens), _stream(A)))
    return 90


def exl3_mgemm_act(G: torch.Tensor, U: torch.Tensor, B: torch.Tensor, C: torch.Tensor, suh: torch.Tensor, svh: torch.Tensor, indices, weights,
                   K: int, mcg: int, mul1: int, min_index: int = -1, max_index: int = -1, num_tokens: int = 1) -> int:
    """exl3_mgemm (indexed) whose per-slot input is fp16(silu(G_j) * U_j): G, U [bszm, m, k] fp16 = the gate / up outputs of the routed experts
    (the silu_mul launch of the MoE block folded into the down launch).  Other arguments as exl3_mgemm."""
    _dev(G)
    _req(G.dtype == torch.half and U.dtype == torch.half and G.shape == U.shape and G.dim() == 3 and G.is_contiguous() and U.is_contiguous(),
         "exl3_mgemm_act: G, U must be contiguous float16 [bszm, m, k]")
    _req(B.dtype == torch.long and suh.dtype == torch.long and svh.dtype == torch.long, "exl3_mgemm_act: B, suh, svh must be int64 pointer tensors")
    _req(C.dim() == 3 and C.is_contiguous() and C.shape[0] == G.shape[0] and C.shape[1] == G.shape[1] and C.dtype in (torch.half, torch.float), "exl3_mgemm_act: bad C")
    bszm, m, k = G.shape
    n = C.shape[2]
    _req(k % 128 == 0 and n % 128 == 0, "exl3_mgemm_act: k and n must be divisible by 128")
    _req(num_tokens == 1 or min_index < 0, "exl3_mgemm_act: num_tokens > 1 is not compatible with an expert range")
    _check(_lib.lib().exl3_mgemm_indexed_act(_p(G), _p(U), _p(B), _p(suh), _p(svh), _p(indices), _p(weights), bszm, _p(C), m, k, n, int(K),
                                             _cb(bool(mcg), bool(mul1)), int(C.dtype == torch.float), int(min_index), int(max_index),
                                             int(num_tokens), _stream(G)))
    return 90


def exl3_mgemm_bcast(A: torch.Tensor, Bs: list[torch.Tensor], Cs: list[torch.Tensor], suhs: list[torch.Tensor],
                     svhs: list[torch.Tensor], mcg: bool = False, mul1: bool = False, force_split: int = 0) -> int:
    """Broadcast form of quant/exl3_gemm.cuh:58-78 (indices == None): one A against several matrices in ONE launch
    (fused q/k/v, gate/up).  The reference passes int64 tensors of device pointers built from these same tensors
    (modules/multilinear.py:30-32)."""
    _dev(A)
    cnt = len(Bs)
    _req(cnt == len(Cs) == len(suhs) == len(svhs) and 1 <= cnt <= 4, "exl3_mgemm: between 1 and 4 matrices")
    m = k = K = None
    ns = []
    for B, C, su, sv in zip(Bs, Cs, suhs, svhs):
        mm, kk, nn, KK = _gemm_checks(A, B, C, su, sv)
        _req(m is None or (mm == m and kk == k and KK == K), "exl3_mgemm: all matrices must share m, k and K")
        _req(C.dtype == Cs[0].dtype, "exl3_mgemm: all outputs must share a dtype")
        m, k, K = mm, kk, KK
        ns.append(nn)
    arr = lambda ts: (_vp * cnt)(*[t.data_ptr() for t in ts])
    return _check(_lib.lib().exl3_mgemm(_p(A), arr(Bs), arr(Cs), arr(suhs), arr(svhs), (ctypes.c_int * cnt)(*ns), cnt,
                                        m, k, K, _cb(mcg, mul1), int(Cs[0].dtype == torch.float), int(force_split), _stream(A)))


MOE_ACT_SILU, MOE_ACT_GELU, MOE_ACT_RELU2_NOGATE = 0, 1, 2           # quant/exl3_moe_common.cuh:6-8
MOE_SLOT_ROWS = 16                                                    # rows of one indexed-launch slot


def exl3_moe_max_concurrency(device: int) -> int:
    """quant/exl3_moe.cu:14-18 (SMs / SMs-per-expert): how many experts the caller should size its temp buffers for.  This build needs no
    temp buffers (the arguments are accepted and ignored); the value only has to be positive."""
    return 32


def exl3_moe(hidden_state, output_state, expert_count, token_sorted, weight_sorted, temp_state_g, temp_state_u, temp_intermediate_g,
             temp_intermediate_u, act_function: int, K_gate: int, K_up: int, K_down: int, gate_ptrs_trellis, gate_ptrs_suh, gate_ptrs_svh,
             up_ptrs_trellis, up_ptrs_suh, up_ptrs_svh, down_ptrs_trellis, down_ptrs_suh, down_ptrs_svh, gate_mcg: bool, gate_mul1: bool,
             up_mcg: bool, up_mul1: bool, down_mcg: bool, down_mul1: bool, act_limit: float, num_active: int):
    """quant/exl3_moe.cu:99-301 (same argument list): output_state (bsz, hidden) fp32 += weight * down_e(silu(gate_e(x)) * up_e(x)) for every
    (token, expert) assignment of an expert e with 0 < expert_count[e] <= max_tokens_per_expert (= temp_state_g.shape[1]); busier experts are left
    to the caller's other route, as in the reference.  token_sorted / weight_sorted list the assignments grouped by expert.
    The reference runs this as ONE persistent cooperative kernel (expert tickets, grid-wide hand-offs between gate|up, activation and down); on
    MI355X an in-kernel grid-wide hand-off costs more than a kernel boundary (profiles/NOTES.md B 4.7 e), so the same work is three indexed launches --
    gate and up of all (expert, <= 16-row chunk) slots, then down with silu * mul formed in its prologue -- and a weighted scatter-add.
    The slot list is built on the DEVICE (exl3_moe_build_slots: prefix sum over expert_count, <= 16-row chunks, over-limit experts masked), every
    shape depends on tensor sizes only and the weighted scatter runs in a fixed order: the op has no host round trip, can be captured in a hipGraph
    and replays bit for bit.  Gated SiLU without a limit takes the route above; GELU, the non-gated relu^2 (MOE_ACT_RELU2_NOGATE: no gate GEMM, the
    gate lane is relu(up): hadamard_inner.cuh:343-378) and any act_limit (up clamped to +-limit, the activated gate to <= limit: :389-397) form
    act(g) * u with one act_mul launch between the up and the down launches."""
    _dev(hidden_state)
    if num_active == 0:
        return
    _req(act_function in (MOE_ACT_SILU, MOE_ACT_GELU, MOE_ACT_RELU2_NOGATE), "exl3_moe: activation must be SiLU, GELU or the non-gated relu^2")
    _req(float(act_limit) >= 0.0, "exl3_moe: act_limit must be >= 0")
    _req(hidden_state.dtype == torch.half and hidden_state.dim() == 2 and hidden_state.is_contiguous(), "exl3_moe: hidden_state must be contiguous float16 (bsz, hidden)")
    _req(output_state.dtype == torch.float and output_state.shape == hidden_state.shape, "exl3_moe: output_state must be float32 with hidden_state's shape")
    _req(expert_count.dtype == torch.long and expert_count.dim() == 1 and token_sorted.dtype == torch.long and token_sorted.dim() == 1,
         "exl3_moe: expert_count / token_sorted must be 1-D int64")
    _req(weight_sorted.shape == token_sorted.shape and weight_sorted.dtype == torch.half, "exl3_moe: weight_sorted must be float16 with token_sorted's shape")
    _req(temp_state_g.dim() == 3 and temp_intermediate_g.dim() == 3, "exl3_moe: temp buffers must be 3-D")
    bsz, hidden = hidden_state.shape
    inter = temp_intermediate_g.shape[2]
    max_rows = temp_state_g.shape[1]
    E = expert_count.shape[0] - 1
    for t in (gate_ptrs_trellis, gate_ptrs_suh, gate_ptrs_svh, up_ptrs_trellis, up_ptrs_suh, up_ptrs_svh, down_ptrs_trellis, down_ptrs_suh, down_ptrs_svh):
        _req(t.dtype == torch.long and t.dim() == 1 and t.shape[0] >= E, "exl3_moe: pointer tables must be int64 with one entry per expert")
    T = token_sorted.shape[0]
    if T == 0:
        return
    # Every shape below depends on tensor SIZES only (no expert_count.tolist(), no boolean-mask indexing): the op is capturable in a hipGraph.
    # The slot list -- one slot = up to m consecutive assignments of one accepted expert -- is built on the device (exl3_moe_build_slots); unused
    # slots carry expert -1 and are skipped by the indexed launches; padded rows repeat an assignment and are never scattered (rowmap).
    # Slot height: a token routes to DISTINCT experts, so an expert receives at most bsz rows -- at decode sizes (bsz 1..4) the slots are 1..4 rows and
    # take the generation-4 launches; padding every slot to min(16, T) rows gathered up to top_k x the rows that exist.  (m is only a chunk size: an
    # expert with more rows than m gets several slots, ns bounds their number.)
    _req(max_rows >= 1, "exl3_moe: temp_state_g must hold at least one row per expert (max_tokens_per_expert >= 1)")
    m = max(1, min(MOE_SLOT_ROWS, max_rows, T, bsz))
    ns = min(E, T) + T // m
    dev = hidden_state.device
    slot_expert = torch.empty((ns,), dtype=torch.long, device=dev)
    slot_tok = torch.empty((ns * m,), dtype=torch.long, device=dev)
    rowmap = torch.empty((T,), dtype=torch.int32, device=dev)
    st = _stream(hidden_state)
    _check(_lib.lib().exl3_moe_build_slots(_p(expert_count), _p(token_sorted), E, T, max_rows, m, ns, _p(slot_expert), _p(slot_tok), _p(rowmap), st))
    A = hidden_state.index_select(0, slot_tok).view(ns, m, hidden)
    gated = act_function != MOE_ACT_RELU2_NOGATE
    U = torch.empty((ns, m, inter), dtype=torch.half, device=dev)
    G = torch.empty_like(U) if gated else None
    if gated:
        exl3_mgemm(A, gate_ptrs_trellis, G, gate_ptrs_suh, None, gate_ptrs_svh, slot_expert, None, K_gate, -1, gate_mcg, gate_mul1, -1, -1, 0)
    exl3_mgemm(A, up_ptrs_trellis, U, up_ptrs_suh, None, up_ptrs_svh, slot_expert, None, K_up, -1, up_mcg, up_mul1, -1, -1, 0)
    D = torch.empty((ns, m, hidden), dtype=torch.float, device=dev)
    if act_function == MOE_ACT_SILU and float(act_limit) == 0.0:
        exl3_mgemm_act(G, U, down_ptrs_trellis, D, down_ptrs_suh, down_ptrs_svh, slot_expert, None, K_down, down_mcg, down_mul1)
    else:
        # (skipped slots hold uninitialised rows: act_mul runs over them, the down launch and the scatter never read them)
        Y = torch.empty_like(U)
        act_mul(G if gated else U, U, Y, {MOE_ACT_SILU: ACT_SILU, MOE_ACT_GELU: ACT_GELU, MOE_ACT_RELU2_NOGATE: ACT_RELU}[act_function], float(act_limit))
        exl3_mgemm(Y, down_ptrs_trellis, D, down_ptrs_suh, None, down_ptrs_svh, slot_expert, None, K_down, -1, down_mcg, down_mul1, -1, -1, 0)
    # weighted scatter in a fixed order per token (bit-reproducible; the round-2 index_add_ was an atomic scatter)
    _check(_lib.lib().exl3_moe_scatter(_p(D), _p(rowmap), _p(token_sorted), _p(weight_sorted), _p(output_state), bsz, T, hidden, st))


def moe_scatter(D: torch.Tensor, rowmap: torch.Tensor, token_sorted: torch.Tensor, weight_sorted: torch.Tensor, out: torch.Tensor):
    """out[t] (fp32, += in place) += sum over token t's assignments p, in ascending p, of weight_sorted[p] * D[rowmap[p]] -- the weighted scatter of a MoE
    block in a FIXED order (bit-reproducible; the reference's index_add_ is an atomic scatter, modules/block_sparse_mlp.py:1308-1309).  rowmap[p] < 0: skipped."""
    _dev(D)
    _req(D.dtype == torch.float and out.dtype == torch.float and D.dim() == 2 and out.dim() == 2 and D.shape[1] == out.shape[1] and D.is_contiguous() and out.is_contiguous(),
         "moe_scatter: D / out must be contiguous float32 (rows, hidden)")
    _req(rowmap.dtype == torch.int32 and token_sorted.dtype == torch.long and weight_sorted.dtype == torch.half and rowmap.numel() == token_sorted.numel() == weight_sorted.numel(),
         "moe_scatter: rowmap int32, token_sorted int64, weight_sorted float16, one entry per assignment")
    _req(out.shape[1] % 4 == 0, "moe_scatter: hidden must be a multiple of 4")
    _check(_lib.lib().exl3_moe_scatter(_p(D), _p(rowmap), _p(token_sorted), _p(weight_sorted), _p(out), out.shape[0], token_sorted.numel(), out.shape[1], _stream(D)))


def hgemm(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor):
    """hgemm.cu:19-102: c = a @ b, fp16 inputs, fp32 accumulate, c fp16/fp32 (may be a column slice)."""
    _dev(a)
    _req(a.dtype == torch.half and b.dtype == torch.half, "hgemm: a and b must be float16")
    _req(c.dtype in (torch.half, torch.float), "hgemm: c must be float16 or float32")
    _req(a.dim() == 2 and b.dim() == 2 and c.dim() == 2, "hgemm: tensors must be 2-D")
    _req(a.shape[1] == b.shape[0] and a.shape[0] == c.shape[0] and b.shape[1] == c.shape[1], "hgemm: shape mismatch")
    _req(a.is_contiguous() and b.is_contiguous() and c.stride(1) == 1, "hgemm: a, b contiguous; c unit column stride")
    _check(_lib.lib().exl3_hgemm(_p(a), _p(b), _p(c), a.shape[0], a.shape[1], b.shape[1], c.stride(0),
                                 int(c.dtype == torch.float), _stream(a)))


#: The prefill GEMMs (hgemm_nt) run the hand-written NT MFMA kernel (exl3_gemm_nt2.hip: assembly K-loop, one wave per SIMD) wherever its tiles apply; the
#: library GEMM (hipBLASLt) takes the other shapes, and all of them with EXL3_HIP_GEMM_NT=0.  On the four Llama-3.1-8B prefill shapes the own kernel measures
#: 1.03 / 0.97 / 0.95 / 0.98 x the library (q|k|v, o, gate|up, down; profiles/r06_gemm_nt_vs_hipblaslt.json), 0.96 x over the whole chunk.
_GEMM_NT_OWN = __import__("os").environ.get("EXL3_HIP_GEMM_NT", "1") == "1"


def hgemm_nt(a: torch.Tensor, bt: torch.Tensor, c: torch.Tensor, accumulate: bool = False, route: str | None = None):
    """c = a @ bt.T (+ c if accumulate): bt is B^T, (n, k) with unit column stride (row stride >= k) -- both operands K-major.
    route: None = the module default (own kernel where its tiles apply, EXL3_HIP_GEMM_NT=0: library), "library" / "own" force one (tests, A/B tools)."""
    _dev(a)
    _req(a.dtype == torch.half and bt.dtype == torch.half and c.dtype in (torch.half, torch.float), "hgemm_nt: bad dtypes")
    _req(a.dim() == 2 and bt.dim() == 2 and c.dim() == 2, "hgemm_nt: tensors must be 2-D")
    _req(a.shape[1] == bt.shape[1] and a.shape[0] == c.shape[0] and bt.shape[0] == c.shape[1], "hgemm_nt: shape mismatch")
    _req(a.stride(1) == 1 and bt.stride(1) == 1 and c.stride(1) == 1, "hgemm_nt: a, bt, c need unit column stride")
    _req(not accumulate or c.dtype == torch.half, "hgemm_nt: accumulate needs a float16 c")
    if (_GEMM_NT_OWN if route is None else route == "own") and c.dtype == torch.half and bt.shape[0] % 128 == 0 and a.shape[1] % 64 == 0 and a.stride(0) % 8 == 0 and bt.stride(0) % 8 == 0 \
            and c.stride(0) % 8 == 0 and a.shape[0] >= 256 and (a.data_ptr() | bt.data_ptr() | c.data_ptr()) % 16 == 0 \
            and (route == "own" or gemm_nt_own_fills_chip(a.shape[0], bt.shape[0], a.device)):
        # the hand-written MFMA GEMM (exl3_gemm_nt2.hip) wherever its tile shapes apply (k % 64, n % 128, rows >= one tile)
        gemm_nt_mfma(a, bt, c, 1 if accumulate else 0)
        return
    if a.is_contiguous():
        _check(_lib.lib().exl3_hgemm_nt(_p(a), _p(bt), _p(c), a.shape[0], a.shape[1], bt.shape[0], bt.stride(0), c.stride(0),
                                        int(c.dtype == torch.float), int(accumulate), _stream(a)))
    else:
        # a is a column range of a wider row-major matrix
        _check(_lib.lib().exl3_hgemm_nt_lda(_p(a), a.stride(0), _p(bt), _p(c), a.shape[0], a.shape[1], bt.shape[0], bt.stride(0), c.stride(0),
                                            int(c.dtype == torch.float), int(accumulate), _stream(a)))


def gemm_nt_grouped(a: torch.Tensor, bt: torch.Tensor, c: torch.Tensor, rows: torch.Tensor, epi: int = 0):
    """Grouped form of the hand-written NT GEMM (exl3_gemm_nt2_grouped): bt (count, n, k) fp16 holds one W^T per problem, a (R, k) / c (R, n) -- (R, n / 2) for epi 2,
    float32 for epi 3 -- the problems' rows back to back, rows (count + 1,) int32 ON THE DEVICE their boundaries (rows[count] <= R; never read on the host)."""
    _dev(a)
    _req(a.dtype == torch.half and bt.dtype == torch.half and a.dim() == 2 and bt.dim() == 3 and c.dim() == 2, "gemm_nt_grouped: a (R, k), bt (count, n, k) float16")
    _req(c.dtype == (torch.float if epi == 3 else torch.half), "gemm_nt_grouped: c float16 (float32 for epi 3)")
    _req(rows.dtype == torch.int32 and rows.is_cuda and rows.is_contiguous() and rows.numel() == bt.shape[0] + 1, "gemm_nt_grouped: rows = device int32 (count + 1)")
    _req(a.stride(1) == 1 and bt.stride(2) == 1 and c.stride(1) == 1 and a.shape[1] == bt.shape[2] and c.shape[0] == a.shape[0], "gemm_nt_grouped: unit column strides, shared k, shared rows")
    _req(c.shape[1] == (bt.shape[1] // 2 if epi == 2 else bt.shape[1]), "gemm_nt_grouped: output columns")
    _check(_lib.lib().exl3_gemm_nt2_grouped(_p(a), a.stride(0), _p(bt), bt.stride(1), bt.stride(0), _p(c), c.stride(0), _p(rows), bt.shape[0], a.shape[0],
                                            a.shape[1], bt.shape[1], int(epi), _stream(a)))


_CUS = {}


def gemm_nt_own_fills_chip(m: int, n: int, device=None, wide_only: bool = False) -> bool:
    """Whether the own NT GEMM is the right kernel for an (m, n) problem: one workgroup per CU and tile, so a problem of few tiles leaves CUs idle where the library's
    stream-K kernel splits k.  Measured (MI355X, 8B shapes, rows 256 .. 4096): within 0.9 - 1.1 x the library where the tile form it picks (256 x 256, or 256 x 128 when that
    wastes fewer rounds) fills >= 70 % of its rounds, 0.66 - 0.85 x below (rows 512 - 1024 on the 4096-column projections).  Same tile choice as exl3_gemm_nt2_mfma."""
    dev = torch.cuda.current_device() if device is None else (device.index if device.index is not None else torch.cuda.current_device())
    cus = _CUS.get(dev)
    if cus is None:
        cus = _CUS[dev] = max(1, torch.cuda.get_device_properties(dev).multi_processor_count)
    tm = (m + 255) // 256
    tw, tn = tm * (n // 256), tm * (n // 128)
    rw, rn = -(-tw // cus), -(-tn // cus)
    narrow = (not wide_only) and (n % 256 != 0 or 0.54 * rn < rw)
    tiles, rounds = (tn, rn) if narrow else (tw, rw)
    return tiles >= 0.7 * rounds * cus


def gemm_nt_own_default() -> bool:
    """Whether ext.hgemm_nt's default route is the hand-written kernel (EXL3_HIP_GEMM_NT != 0)."""
    return _GEMM_NT_OWN


def gemm_nt_mfma(a: torch.Tensor, bt: torch.Tensor, c: torch.Tensor, epi: int = 0, gen: int = 0):
    """The hand-written NT MFMA GEMM: c = a @ bt.T; epi 0 store, 1 c += (fp16 residual add), 2 silu(gate) * up on 128 | 128
    row pairs of bt (c has bt.shape[0] / 2 columns).  gen 0: exl3_gemm_nt2.hip (one wave per SIMD, assembly K-loop; k % 64, n % 128) with the tile it picks for the
    shape; gen 3 / 4 force its 256 x 256 / 256 x 128 tile, gen 2 its first version (32-deep K-tiles, k % 128); gen 1: exl3_gemm_nt.hip (k % 64, n % 256).
    Raises for shapes outside the chosen kernel."""
    _dev(a)
    _req(a.dtype == torch.half and bt.dtype == torch.half and c.dtype == torch.half and a.dim() == 2 and bt.dim() == 2 and c.dim() == 2, "gemm_nt_mfma: 2-D float16 tensors")
    _req(a.stride(1) == 1 and bt.stride(1) == 1 and c.stride(1) == 1 and a.shape[1] == bt.shape[1], "gemm_nt_mfma: unit column strides, shared k")
    _req(c.shape[0] == a.shape[0] and c.shape[1] == (bt.shape[0] // 2 if epi == 2 else bt.shape[0]), "gemm_nt_mfma: output shape")
    if gen == 0 and a.shape[1] % 64 != 0:
        gen = 1
    fn = _lib.lib().exl3_gemm_nt2_mfma if gen != 1 else _lib.lib().exl3_gemm_nt_mfma
    _check(fn(_p(a), a.stride(0), _p(bt), bt.stride(0), _p(c), c.stride(0), a.shape[0], a.shape[1], bt.shape[0], int(epi) | {2: 0x100, 3: 0x400, 4: 0x200, 5: 0x800, 6: 0xC00, 7: 0x1400}.get(gen, 0), _stream(a)))


def reconstruct_had_slice_t(unpacked_t: torch.Tensor, packed: torch.Tensor, suh: torch.Tensor, svh: torch.Tensor,
                            K: int, mcg: bool, mul1: bool, n_offset: int):
    """reconstruct_had_slice written transposed: unpacked_t is (n_size, k) with unit column stride (row stride >= k) and receives W^T."""
    _dev(unpacked_t)
    _req(packed.dim() == 3 and packed.dtype == torch.int16 and packed.shape[2] == 16 * K, "packed must be a 3-D int16 tensor with dim 2 = 16*K")
    _req(unpacked_t.dtype == torch.half and unpacked_t.dim() == 2 and unpacked_t.shape[1] == packed.shape[0] * 16, "unpacked_t must be (n, k) float16")
    _req(unpacked_t.stride(1) == 1 and packed.is_contiguous(), "unpacked_t needs a unit column stride; packed contiguous")
    _req(suh.dtype == torch.half and svh.dtype == torch.half, "suh/svh must be float16")
    _req(suh.numel() >= unpacked_t.shape[1] and svh.numel() >= unpacked_t.shape[0], "suh/svh too small")
    _check(_lib.lib().exl3_reconstruct_had_t(_p(unpacked_t), unpacked_t.stride(0), _p(packed), _p(suh), _p(svh), packed.shape[0], packed.shape[1],
                                             K, _cb(mcg, mul1), n_offset, unpacked_t.shape[0], _stream(unpacked_t)))


def reconstruct_had_multi_t(unpacked_t: torch.Tensor, packed: list, suhs: list, svhs: list, K: int, mcg: bool, mul1: bool, interleave: bool = False):
    """W^T of up to 4 whole matrices (same k, K, codebook) stacked along n into unpacked_t (sum n_i, k) in ONE launch.  interleave: the matrices' 128-row blocks
    alternate (block j of matrix i at rows (j * count + i) * 128; equal n): the operand of gemm_nt_mfma's fused silu(gate) * up epilogue."""
    _dev(unpacked_t)
    cnt = len(packed)
    _req(1 <= cnt <= 4 and len(suhs) == cnt and len(svhs) == cnt, "reconstruct_had_multi_t: between 1 and 4 matrices")
    k = packed[0].shape[0] * 16
    _req(all(p.dim() == 3 and p.dtype == torch.int16 and p.shape[2] == 16 * K and p.shape[0] * 16 == k and p.is_contiguous() for p in packed), "packed: 3-D int16, dim 2 = 16*K, shared k")
    _req(unpacked_t.dtype == torch.half and unpacked_t.dim() == 2 and unpacked_t.shape[1] == k and unpacked_t.stride(1) == 1, "unpacked_t must be (sum n, k) float16")
    _req(unpacked_t.shape[0] == sum(p.shape[1] * 16 for p in packed), "unpacked_t rows = sum of the matrices' n")
    _req(all(t.dtype == torch.half for t in suhs + svhs), "suh/svh must be float16")
    tn = (ctypes.c_int * cnt)(*[p.shape[1] for p in packed])
    _req(not interleave or all(p.shape[1] == packed[0].shape[1] for p in packed), "reconstruct_had_multi_t (interleave): matrices of one n")
    fn = _lib.lib().exl3_reconstruct_had_multi_t_interleaved if interleave else _lib.lib().exl3_reconstruct_had_multi_t
    _check(fn(_p(unpacked_t), unpacked_t.stride(0), _parr(packed), _parr(suhs), _parr(svhs), tn, cnt, packed[0].shape[0], int(K), _cb(mcg, mul1), _stream(unpacked_t)))


def hgemm_acc(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor):
    """c (fp16, in place) = fp16(a @ b + c): hgemm + residual add in the GEMM epilogue (one rounding, as fp32 output + `x += y`)."""
    _dev(a)
    _req(a.dtype == torch.half and b.dtype == torch.half and c.dtype == torch.half, "hgemm_acc: tensors must be float16")
    _req(a.dim() == 2 and b.dim() == 2 and c.dim() == 2, "hgemm_acc: tensors must be 2-D")
    _req(a.shape[1] == b.shape[0] and a.shape[0] == c.shape[0] and b.shape[1] == c.shape[1], "hgemm_acc: shape mismatch")
    _req(a.is_contiguous() and b.is_contiguous() and c.stride(1) == 1, "hgemm_acc: a, b contiguous; c unit column stride")
    _check(_lib.lib().exl3_hgemm_acc(_p(a), _p(b), _p(c), a.shape[0], a.shape[1], b.shape[1], c.stride(0), _stream(a)))


class BC_LinearEXL3:
    """libtorch/linear.h:29-64, linear.cpp:34-71: holder of {trellis, suh, svh, K, bias, mcg, mul1, xh}."""

    def __init__(self, trellis, suh, svh, K, bias, mcg, mul1, xh):
        self.trellis, self.suh, self.svh, self.K = trellis, suh, svh, int(K)
        self.bias, self.mcg, self.mul1, self.xh = bias, bool(mcg), bool(mul1), xh

    def run(self, x: torch.Tensor, y: torch.Tensor):
        exl3_gemm(x, self.trellis, y, self.suh, self.xh, self.svh, -1, self.mcg, self.mul1, 0, bias=self.bias)

    def run_alloc(self, x: torch.Tensor, out_features: int, output_fp32: bool) -> torch.Tensor:
        y = torch.empty(x.shape[:-1] + (out_features,), dtype=torch.float if output_fp32 else torch.half, device=x.device)
        self.run(x, y)
        return y


class BC_LinearFP16:
    """libtorch/linear.h:10-27, linear.cpp:11-25: unquantized fp16 Linear holder {weight (k, n), bias}: y = x @ weight (+ bias) through hgemm.
    (The reference takes at::matmul_out when dtypes match and hgemm otherwise; both are the library GEMM.)"""

    def __init__(self, weight, bias=None):
        self.weight, self.bias = weight, bias

    def run(self, x: torch.Tensor, y: torch.Tensor):
        x2 = x.view(-1, x.shape[-1])
        hgemm(x2, self.weight, y.view(x2.shape[0], -1))
        if self.bias is not None:
            y2 = y.view(x2.shape[0], -1)
            add(y2, self.bias.view(1, -1).expand(x2.shape[0], -1).contiguous() if x2.shape[0] > 1 else self.bias)


class BC_Attention:
    """libtorch/attention.h:24-228, attention.cpp:246-504: the decode attention block of one layer as one runner -- q / k / v (/ gate) projections,
    head norms + RoPE, append to the paged cache, attention over the cache, output gate, o_proj.  Same constructor arguments and run() signature as
    the reference's class; q_len 1 .. 16 new tokens per sequence, bsz <= 8, head_dim 64 / 128, page size 256.
    Quantized cache (`quant_cache`, 2-8 bits): flash-decoding attention straight from the quantized pages, learned attention sinks in its merge
    kernel.  fp16 cache: rows appended by index, causal attention of the chunk over the fp16 pages (the prefill path's flash-attention kernel; no
    sinks).  V norm, K-as-V and the rope options (llama-4 query scale, norm after the rotation, `rotate_dims`) are composed from the ops of this
    module as the reference composes them (attention.cpp:335-395); a padded hidden dim is staged through zero-padded statics; the output gates
    (headwise / full / interleaved, sigmoid or softplus: attention.cpp:283-333, 483-494) are the reference's own op sequence over this module's ops.
    The reference's slot machinery exists to hold AOT-compiled Triton kernels and their statics: needs_configure() is always False here and
    configure_slot() accepts and ignores its arguments; capture the whole decode step in one hipGraph instead (all launches of run() are
    capturable, the per-call tensors are read by pointer)."""

    MAX_BSZ = 8
    MAX_QLEN = 16

    def __init__(self, num_q_heads, num_kv_heads, head_dim, hidden_size, hidden_size_padded, page_size, q_proj, k_proj, v_proj,
                 kv_ptrs_trellis=None, kv_ptrs_suh=None, kv_ptrs_svh=None, kv_K=0, kv_mcg=False, kv_mul1=False, o_proj=None,
                 use_k_as_v=False, gate_mode=0, gate_softplus=False, g_proj=None, g_weight=None, qg_ptrs_trellis=None, qg_ptrs_suh=None,
                 qg_ptrs_svh=None, qg_K=0, qg_mcg=False, qg_mul1=False, q_norm=None, k_norm=None, norm_eps=1e-6, norm_constant_bias=0.0,
                 v_norm=False, v_norm_w=None, v_norm_eps=1e-6, v_norm_constant_bias=0.0, v_norm_constant_scale=1.0, inv_freq=None,
                 rope_style=2, attn_factor=1.0, l4_scaling_beta=0.0, l4_scaling_original=0, post_rope_norm=False, rotate_dims=1,
                 quant_cache=True, cache_k=None, cache_v=None, cache_k_scales=None, cache_v_scales=None, xh=None, h32=None, sinks=None):
        _req(not quant_cache or (cache_k_scales is not None and cache_v_scales is not None), "BC_Attention: a quantized cache needs its scale tensors")
        _req(quant_cache or (cache_k is not None and cache_k.dtype == torch.half and cache_k.dim() == 4 and cache_k.shape[1] == page_size
                             and cache_k.shape[2] == num_kv_heads and cache_k.shape[3] == head_dim and cache_v is not None and cache_v.shape == cache_k.shape),
             "BC_Attention: an unquantized cache is float16 (pages, page_size, kv_heads, head_dim)")
        _req(quant_cache or sinks is None, "BC_Attention: attention sinks need the quantized cache (the merge kernel of the quant-cache-direct attention)")
        self.quant_cache = bool(quant_cache)
        _req(gate_mode in (0, 1, 2, 3), "BC_Attention: gate_mode must be 0 (none), 1 (headwise), 2 (full) or 3 (interleaved)")
        _req(gate_mode != 1 or g_weight is not None, "BC_Attention: headwise gate requires the fp16 gate weight")                        # attention.cpp:303
        _req(gate_mode != 2 or g_weight is not None or g_proj is not None or qg_ptrs_trellis is not None,
             "BC_Attention: full gate without fused qg needs a g projection")                                                          # attention.cpp:317
        _req(g_weight is None or (g_weight.dtype == torch.half and g_weight.dim() == 2 and g_weight.shape[0] == hidden_size
                                  and g_weight.shape[1] == (num_q_heads if gate_mode == 1 else num_q_heads * head_dim)),
             "BC_Attention: g_weight must be float16 (hidden, heads) for the headwise gate, (hidden, heads * head_dim) for the full gate")
        _req(hidden_size_padded >= hidden_size and hidden_size_padded % 128 == 0, "BC_Attention: hidden_size_padded must be a multiple of 128 >= hidden_size")
        _req(not use_k_as_v or k_proj is not None, "BC_Attention: K-as-V needs the separate k projection")
        _req(head_dim in (64, 128) and num_q_heads % num_kv_heads == 0, "BC_Attention: head_dim must be 64 or 128")
        _req(page_size == 256, "BC_Attention: page size must be 256")
        _req(q_proj is not None and o_proj is not None and (use_k_as_v or (k_proj is not None and v_proj is not None) or kv_ptrs_trellis is not None),
             "BC_Attention: no k/v projection path")
        _req(not v_norm or v_norm_w is None or v_norm_w.numel() == head_dim, "BC_Attention: v_norm weight must have head_dim entries")
        _req(inv_freq is not None or (q_norm is None and k_norm is None), "BC_Attention: head norms ride on the rope kernel (NoPE modules cannot have them)")
        self.num_q_heads, self.num_kv_heads, self.head_dim, self.hidden_size, self.page_size = num_q_heads, num_kv_heads, head_dim, hidden_size, page_size
        self.hidden_size_padded = int(hidden_size_padded)
        self.q_proj, self.k_proj, self.v_proj, self.o_proj = q_proj, k_proj, v_proj, o_proj
        self.kv_ptrs = (kv_ptrs_trellis, kv_ptrs_suh, kv_ptrs_svh, int(kv_K), bool(kv_mcg), bool(kv_mul1)) if kv_ptrs_trellis is not None else None
        self.q_norm, self.k_norm, self.norm_eps, self.norm_constant_bias = q_norm, k_norm, norm_eps, norm_constant_bias
        self.inv_freq, self.rope_style, self.attn_factor = inv_freq, int(rope_style), float(attn_factor)
        self.l4_beta, self.l4_orig, self.post_rope_norm, self.rotate_dims = float(l4_scaling_beta), int(l4_scaling_original), bool(post_rope_norm), int(rotate_dims)
        self.use_k_as_v, self.v_norm, self.v_norm_w = bool(use_k_as_v), bool(v_norm), v_norm_w
        self.sinks = sinks                                             # float32 [heads_q] or None: the combine step's learned sink logits
        self.gate_mode, self.gate_softplus, self.g_proj, self.g_weight = int(gate_mode), bool(gate_softplus), g_proj, g_weight
        self.qg_ptrs = (qg_ptrs_trellis, qg_ptrs_suh, qg_ptrs_svh, int(qg_K), bool(qg_mcg), bool(qg_mul1)) if qg_ptrs_trellis is not None else None
        self.v_norm_eps, self.v_norm_bias, self.v_norm_scale = float(v_norm_eps), float(v_norm_constant_bias), float(v_norm_constant_scale)
        self.cache_k, self.cache_v, self.cache_k_scales, self.cache_v_scales = cache_k, cache_v, cache_k_scales, cache_v_scales
        self.xh = xh
        self._st = {}

    def needs_configure(self, bsz: int, q_len: int) -> bool:
        return False

    def configure_slot(self, *args, **kwargs):
        return None

    def _statics(self, bsz: int, q_len: int, pages_per_seq: int, dev):
        key = (bsz, q_len, pages_per_seq)
        st = self._st.get(key)
        if st is None:
            hq, hkv, hd = self.num_q_heads, self.num_kv_heads, self.head_dim
            max_len = pages_per_seq * self.page_size
            rows = bsz * q_len
            st = {
                "q": torch.empty((bsz, q_len, hq, hd), dtype=torch.half, device=dev),
                "kv": torch.empty((2, rows, hkv * hd), dtype=torch.half, device=dev),
                "o": torch.empty((bsz, q_len, hq, hd), dtype=torch.half, device=dev),
                "lens": torch.empty((bsz,), dtype=torch.int32, device=dev),
                "xh": torch.empty((2, rows, self.hidden_size_padded), dtype=torch.half, device=dev),
                "max_len": max_len,
            }
            if self.hidden_size_padded != self.hidden_size:
                # attention.cpp:268-280, 497-508: the input is staged through a zero-padded static (the projections' K is the padded width), o_proj
                # writes the padded width and the exact-width columns are copied out
                st["xp"] = torch.zeros((rows, self.hidden_size_padded), dtype=torch.half, device=dev)
                st["yp"] = torch.empty((rows, self.hidden_size_padded), dtype=torch.half, device=dev)
            if self.gate_mode == 1:
                st["g"] = torch.empty((rows, hq), dtype=torch.half, device=dev)
            elif self.gate_mode == 2:
                # q and g as the two halves of one (2, rows, hq * hd) buffer: the fused q + g launch writes both (attention.cpp:291-297)
                st["qg"] = torch.empty((2, rows, hq * hd), dtype=torch.half, device=dev)
                st["q"] = st["qg"][0].view(bsz, q_len, hq, hd)
                st["g"] = st["qg"][1]
            elif self.gate_mode == 3:
                st["qg_i"] = torch.empty((rows, 2 * hq * hd), dtype=torch.half, device=dev)
                st["g"] = torch.empty((rows, hq * hd), dtype=torch.half, device=dev)
            # 2 <= q_len <= 16: token t of sequence b attends as its own one-token sequence (b, t) with the same block-table row and length
            # cache_seqlens[b] + t + 1 -- causal by construction, straight from the quantized pages (no fp16 page window)
            st["ws"] = torch.empty((rows * (hq * hd // 128) * ((max_len + 31) // 32) * 132,), dtype=torch.float, device=dev)
            if q_len > 1:
                st["row_of"] = torch.arange(bsz, device=dev).repeat_interleave(q_len)
                st["tofs"] = (torch.arange(q_len, dtype=torch.int32, device=dev) + 1).repeat(bsz)
                st["bt_v"] = torch.empty((rows, pages_per_seq), dtype=torch.int32, device=dev)
                st["lens_v"] = torch.empty((rows,), dtype=torch.int32, device=dev)
            self._st[key] = st
        return st

    def run(self, bsz: int, q_len: int, x, y, cache_seqlens, block_table, position: int = 0, positions=None, position_ids=None, inv_freq_override=None):
        """x, y: (bsz, q_len, hidden) fp16; cache_seqlens int32 (bsz): tokens in the cache BEFORE this call (the new tokens are appended from
        that position, attention.cpp:395-400); block_table int32 (bsz, pages); RoPE position of (b, t) = position + t | positions[b] + t |
        position_ids[b][t] as in ext.rope.  Attention runs straight from the quantized cache (flash-decoding); for 2 <= q_len <= MAX_QLEN every
        new token is its own query row with length cache_seqlens[b] + t + 1 (causal within the chunk), after the whole chunk was appended."""
        _req(1 <= q_len <= self.MAX_QLEN, "BC_Attention: q_len out of range")
        _req(1 <= bsz <= self.MAX_BSZ, "BC_Attention: bsz out of range")
        _req(self.inv_freq is not None or inv_freq_override is None, "BC_Attention: inv_freq override on a NoPE module")
        hq, hkv, hd = self.num_q_heads, self.num_kv_heads, self.head_dim
        rows = bsz * q_len
        st = self._statics(bsz, q_len, block_table.shape[1], x.device)
        x2 = x.view(rows, self.hidden_size)
        if "xp" in st:
            st["xp"][:, :self.hidden_size].copy_(x2)
            x2 = st["xp"]
        q2, kv = st["q"].view(rows, hq * hd), st["kv"]
        if self.gate_mode == 3:
            # q_proj emits q and g interleaved per head (attention.cpp:283-289)
            self.q_proj.run(x2, st["qg_i"])
            deinterleave_qg(st["qg_i"], q2, st["g"], hd)
        elif self.gate_mode == 2 and self.qg_ptrs is not None and rows <= 16:
            pt, ps, pv, K, mcg, mul1 = self.qg_ptrs                       # fused q + g launch (attention.cpp:291-297)
            exl3_mgemm(x2.view(1, rows, -1), pt, st["qg"], ps, st["xh"], pv, None, None, K, -1, mcg, mul1, -1, -1, 0)
        else:
            self.q_proj.run(x2, q2)
            if self.gate_mode == 1:
                hgemm(x2, self.g_weight, st["g"])                         # one gate per head (attention.cpp:301-305)
            elif self.gate_mode == 2:
                if self.g_weight is not None:
                    hgemm(x2, self.g_weight, st["g"])
                else:
                    _req(self.g_proj is not None, "BC_Attention: more than 16 rows need the separate g projection")
                    self.g_proj.run(x2, st["g"])
        if self.use_k_as_v:
            # attention.cpp:335-360: V shares the K projection's output, taken before the head norm / RoPE touch K (per-head RMSNorm of it, or a copy)
            self.k_proj.run(x2, kv[0])
            if self.v_norm:
                rms_norm(kv[0].view(rows * hkv, hd), self.v_norm_w, kv[1].view(rows * hkv, hd), self.v_norm_eps, self.v_norm_bias, self.v_norm_scale)
            else:
                kv[1].copy_(kv[0])
        elif self.kv_ptrs is not None and rows <= 16:                     # the pointer-table launch takes at most 16 rows per slot
            pt, ps, pv, K, mcg, mul1 = self.kv_ptrs
            exl3_mgemm(x2.view(1, rows, -1), pt, kv, ps, st["xh"], pv, None, None, K, -1, mcg, mul1, -1, -1, 0)
        else:
            _req(self.k_proj is not None and self.v_proj is not None, "BC_Attention: more than 16 rows need the separate k / v projections")
            self.k_proj.run(x2, kv[0]); self.v_proj.run(x2, kv[1])
        if self.v_norm and not self.use_k_as_v:                           # attention.cpp:379-384: per-head norm of V, in place
            v2h = kv[1].view(rows * hkv, hd)
            rms_norm(v2h, self.v_norm_w, v2h, self.v_norm_eps, self.v_norm_bias, self.v_norm_scale)
        k4, v4 = kv[0].view(bsz, q_len, hkv, hd), kv[1].view(bsz, q_len, hkv, hd)
        if self.inv_freq is not None:
            ivf = inv_freq_override if inv_freq_override is not None else self.inv_freq
            rope(st["q"], st["q"], k4, k4, ivf, int(position), positions, position_ids, self.rope_style, self.attn_factor,
                 self.q_norm, self.k_norm, self.norm_eps, self.norm_constant_bias, self.l4_beta, max(self.l4_orig, 1), self.post_rope_norm,
                 self.rotate_dims, 0)
        if not self.quant_cache:
            # fp16 paged cache (attention.cpp:400-418 k_update, then the same attention over fp16 pages): rows appended by one launch,
            # causal attention of the chunk over the pages -- the flash-attention kernel of the prefill path, which a one-token chunk reduces to decode attention
            paged_kv_cache_update(k4, v4, self.cache_k, self.cache_v, block_table, cache_seqlens)
            torch.add(cache_seqlens, q_len, out=st["lens"])
            attn_prefill_paged(st["q"], st["o"], self.cache_k, self.cache_v, block_table, st["lens"])
            self._finish(st, bsz, q_len, rows, y)
            return
        quant_cache_paged(k4.view(bsz, q_len, -1), self.cache_k, self.cache_k_scales, v4.view(bsz, q_len, -1), self.cache_v, self.cache_v_scales,
                          cache_seqlens, block_table, self.page_size, q_len)
        if q_len == 1:
            torch.add(cache_seqlens, 1, out=st["lens"])                   # the attention kernel's lengths include the appended token
            bt_v, lens_v = block_table, st["lens"]
        else:
            torch.index_select(block_table, 0, st["row_of"], out=st["bt_v"])
            torch.index_select(cache_seqlens, 0, st["row_of"], out=st["lens_v"])
            st["lens_v"].add_(st["tofs"])
            bt_v, lens_v = st["bt_v"], st["lens_v"]
        attn_decode_qcache(st["q"].view(rows, hq, hd), st["o"].view(rows, hq, hd), self.cache_k, self.cache_k_scales, self.cache_v, self.cache_v_scales,
                           bt_v, lens_v, st["max_len"], workspace=st["ws"], sinks=self.sinks)
        self._finish(st, bsz, q_len, rows, y)

    def _finish(self, st, bsz: int, q_len: int, rows: int, y):
        """output gate + o_proj (attention.cpp:483-508)"""
        hq, hd = self.num_q_heads, self.head_dim
        if self.gate_mode == 1:                                           # attention.cpp:483-494
            g3 = st["g"].view(bsz, q_len, hq)
            if self.gate_softplus: mul_softplus_broadcast_(st["o"], g3)
            else: mul_sigmoid_broadcast_(st["o"], g3)
        elif self.gate_mode in (2, 3):
            mul_sigmoid_(st["o"].view(rows, hq * hd), st["g"])
        if "yp" in st:
            self.o_proj.run(st["o"].view(rows, hq * hd), st["yp"])
            y.view(rows, self.hidden_size).copy_(st["yp"][:, :self.hidden_size])
        else:
            self.o_proj.run(st["o"].view(rows, hq * hd), y.view(rows, self.hidden_size))


def __getattr__(name: str):
    # PEP 562: anything the wider reference stack looks up that is not part of the EXL3 quantized-linear hot path
    raise AttributeError(f"exllamav3_ext (exllamav3_amd build): op '{name}' is outside the EXL3 quantized-linear hot path and is not provided")


# --------------------------------------------------------------------------------------------------
# norm / rope / cache / elementwise
# --------------------------------------------------------------------------------------------------

def _norm(x, w, y, r, eps, constant_bias, constant_scale, mode):
    _dev(x)
    _req(x.dtype in (torch.half, torch.float) and y.dtype in (torch.half, torch.float), "rms_norm: Invalid datatypes for input/output")
    _req(x.shape == y.shape, "rms_norm: x and y shapes differ")
    _req(x.shape[-1] % 4 == 0, "rms_norm: last dimension must be divisible by 4")
    _req(x.is_contiguous() and y.is_contiguous(), "rms_norm: tensors must be contiguous")
    w_bf16 = 0
    if w is not None:
        _req(w.dtype in (torch.half, torch.bfloat16) and w.numel() == x.shape[-1], "rms_norm: bad weight tensor")
        w_bf16 = int(w.dtype == torch.bfloat16)
    r_fp32 = 0
    if r is not None:
        _req(r.shape == x.shape and r.dtype in (torch.half, torch.float) and r.is_contiguous(), "rms_norm: bad residual tensor")
        r_fp32 = int(r.dtype == torch.float)
    dim = x.shape[-1]
    rows = x.numel() // dim
    _check(_lib.lib().exl3_rms_norm(_p(x), _p(w), _p(y), _p(r), float(eps), float(constant_bias), float(constant_scale),
                                    rows, dim, int(x.dtype == torch.float), int(y.dtype == torch.float), r_fp32, w_bf16, mode, _stream(x)))


def rms_norm(x, w, y, epsilon: float, constant_bias: float = 0.0, constant_scale: float = 1.0,
             span_heads: bool = False, add_residual: bool = False):
    """norm.cuh:7-17"""
    if span_heads:
        x = x.view(x.shape[:-2] + (-1,)) if x.dim() > 2 else x
        y = y.view(x.shape)
    _norm(x, w, y, None, epsilon, constant_bias, constant_scale, 1 if add_residual else 0)


def rms_norm_res_in(x, w, y, r, epsilon: float, constant_bias: float = 0.0, constant_scale: float = 1.0):
    """norm.cuh:29-39: r += x; y = norm(r)"""
    _norm(x, w, y, r, epsilon, constant_bias, constant_scale, 2)


def rope(q, out_q, k, out_k, inv_freq, position: int, positions, position_ids, rope_mode: int, attn_factor: float,
         q_norm=None, k_norm=None, norm_eps: float = 1e-6, norm_constant_bias: float = 0.0, l4_beta: float = 0.0,
         l4_orig: int = 1, post_rope_norm: bool = False, rotate_dims: int = 1, rotate_offset: int = 0):
    """rope.cuh:51-72 with the host checks of rope.cu:345-470.  Llama / Mixtral calls (full rotary width, one position per token, fp16 head-norm
    weights, contiguous heads) take the register kernels of exl3_rope; everything else of the argument list -- partial rotary, `rotate_dims` with 3-D
    position ids, a rotate offset, an angle table, NANOCHAT, bf16 norm weights, the norm after the rotation, the llama-4 query scale, strided heads --
    takes exl3_rope_ex."""
    _dev(q)
    _req(q.dim() == 4 and q.dtype == torch.half and q.stride(3) == 1, "rope: q must be float16 (b, s, h, d) with a dense innermost dim")
    _req(out_q.shape == q.shape and out_q.dtype == torch.half and out_q.stride() == q.stride(), "rope: out_q must share q's layout")
    bsz, seq, hq, hd = q.shape
    _req(q.stride(1) == hq * q.stride(2) and q.stride(0) == seq * q.stride(1), "rope: token stride must be heads * head stride")
    hk, ks = 0, 0
    if k is not None:
        _req(k.dim() == 4 and k.dtype == torch.half and k.stride(3) == 1 and k.shape[0] == bsz and k.shape[1] == seq and k.shape[3] == hd,
             "rope: k is incorrect shape")
        _req(out_k is not None and out_k.shape == k.shape and out_k.stride() == k.stride(), "rope: out_k mismatch")
        hk, ks = k.shape[2], k.stride(2)
        _req(k.stride(1) == hk * ks and k.stride(0) == seq * k.stride(1), "rope: token stride must be heads * head stride")
    _req(inv_freq.dtype == torch.float and 1 <= inv_freq.dim() <= 3 and inv_freq.is_contiguous(), "rope: inv_freq must be contiguous float32")
    partial = inv_freq.shape[-1] * 2
    _req(0 < rotate_dims <= 4, "rotate_dims out of range")
    _req(rotate_dims == 1 or hd == partial * rotate_dims, "rotate_dims is inconsistent with inv_freq and head_dim")
    _req(rotate_offset >= 0 and rotate_offset + partial * rotate_dims <= hd, "rotate_offset out of range")
    _req(positions is None or position_ids is None, "rope: invalid arguments (positions and position_ids)")
    ids_stride = 1
    for t in (positions, position_ids):
        _req(t is None or t.dtype in (torch.int32, torch.int), "rope: positions / position_ids must be int32")
    if positions is not None:
        _req(positions.dim() == 1 and positions.shape[0] == bsz, "positions is incorrect shape")
    if position_ids is not None:
        _req(position_ids.is_contiguous(), "position_ids must be contiguous")
        rd = position_ids.dim()
        _req(rd == 2 or (rd == 3 and position_ids.shape[-1] == rotate_dims), "position_ids wrong number of dims")
        _req(position_ids.shape[0] == bsz and position_ids.shape[1] == seq, "position_ids is incorrect shape")
        if rd == 3:
            ids_stride = rotate_dims
    bf16 = False
    if q_norm is not None:
        _req(q_norm.dim() == 1 and q_norm.shape[0] == hd and q_norm.dtype in (torch.half, torch.bfloat16), "q_norm is incorrect size / dtype")
        _req(k is None or (k_norm is not None and k_norm.dtype == q_norm.dtype and k_norm.shape == q_norm.shape), "q_norm and k_norm must be same dtype")
        bf16 = q_norm.dtype == torch.bfloat16
    table = inv_freq.dim() > 1
    plain = (not table and partial == hd and rotate_dims == 1 and rotate_offset == 0 and ids_stride == 1 and rope_mode in (1, 2) and not bf16
             and not post_rope_norm and l4_beta <= 0.0 and q.is_contiguous() and (k is None or k.is_contiguous()))
    if plain:
        _check(_lib.lib().exl3_rope(_p(q), _p(out_q), _p(k), _p(out_k), _p(inv_freq), bsz, seq, hq, hk, hd, int(position),
                                    _p(positions), _p(position_ids), int(rope_mode), float(attn_factor), _p(q_norm), _p(k_norm),
                                    float(norm_eps), float(norm_constant_bias), _stream(q)))
        return
    # rope.cu:388-395: a table of angles [pos][pairs] or [batch][pos][pairs]; the batch stride of a 2-D table is 0 here (one table for every sequence)
    table_stride = inv_freq.shape[-1] * inv_freq.shape[-2] if inv_freq.dim() == 3 else 0
    _check(_lib.lib().exl3_rope_ex(_p(q), _p(out_q), _p(k), _p(out_k), _p(inv_freq), bsz, seq, hq, hk, hd, q.stride(2), ks, partial, int(position),
                                   _p(positions), _p(position_ids), ids_stride, int(rope_mode), float(attn_factor), _p(q_norm), _p(k_norm), int(bf16),
                                   float(norm_eps), float(norm_constant_bias), int(table), int(table_stride), float(l4_beta), int(l4_orig),
                                   int(bool(post_rope_norm)), int(rotate_dims), int(rotate_offset), _stream(q)))


def _kv_bits(packed: torch.Tensor, scales: torch.Tensor) -> int:
    return packed.shape[-1] // scales.shape[-1]


def quant_cache_cont(inp, out, out_scales, compand_a: float = 0.0):
    """cache/q_cache.cuh: contiguous quantization (compand_a > 0: the cubic level compander of cache/lmq.cuh)"""
    _dev(inp)
    _req(inp.dtype == torch.half and inp.is_contiguous() and inp.shape[-1] % 32 == 0, "quant_cache_cont: bad input")
    dim = inp.shape[-1]
    bits = _kv_bits(out, out_scales)
    _req(2 <= bits <= 8, "quant_cache_cont: bits must be in [2, 8]")
    _check(_lib.lib().exl3_quant_cache_cont_ex(_p(inp), _p(out), _p(out_scales), inp.numel() // dim, dim, bits, float(compand_a), _stream(inp)))


def dequant_cache_cont(inp, in_scales, out, compand_a: float = 0.0):
    _dev(inp)
    dim = out.shape[-1]
    bits = _kv_bits(inp, in_scales)
    _req(out.dtype == torch.half and out.is_contiguous() and dim % 32 == 0 and 2 <= bits <= 8, "dequant_cache_cont: bad arguments")
    _check(_lib.lib().exl3_dequant_cache_cont_ex(_p(inp), _p(in_scales), _p(out), out.numel() // dim, dim, bits, float(compand_a), _stream(inp)))


def quant_cache_paged(k_in, k_out, k_scales, v_in, v_out, v_scales, cache_seqlens, block_table, page_size: int, seq_len: int,
                      compand_a: float = 0.0, in_contiguous: bool = True):
    """cache/q_cache.cuh:48-62.  in_contiguous=False: k_in / v_in are the flat fp16 cache (pages, page, dim), read at each new token's own row."""
    _dev(k_in)
    _req(page_size == 256, "quant_cache_paged: page size must be 256")
    _req(cache_seqlens.dtype == torch.int32 and block_table.dtype == torch.int32, "cache_seqlens / block_table must be int32")
    _req(k_in.is_contiguous() and v_in.is_contiguous() and k_in.dtype == torch.half and v_in.dtype == torch.half, "quant_cache_paged: k_in / v_in must be contiguous fp16")
    dim = k_out.shape[-1] // _kv_bits(k_out, k_scales) * 32
    bsz = block_table.shape[0]
    _check(_lib.lib().exl3_quant_cache_paged_ex(_p(k_in), _p(k_out), _p(k_scales), _p(v_in), _p(v_out), _p(v_scales),
                                                _p(cache_seqlens), _p(block_table), bsz, block_table.shape[1], page_size, seq_len, dim,
                                                _kv_bits(k_out, k_scales), _kv_bits(v_out, v_scales), dim, dim, float(compand_a),
                                                int(bool(in_contiguous)), _stream(k_in)))


def quant_cache_paged_strided(k_in, k_out, k_scales, v_in, v_out, v_scales, cache_seqlens, block_table, page_size: int, seq_len: int):
    """quant_cache_paged on k_in / v_in that are column ranges (bsz * seq_len, dim) of a wider row-major fp16 matrix (unit column stride):
    the fused q|k|v prefill GEMM output, consumed without a split copy."""
    _dev(k_in)
    _req(page_size == 256, "quant_cache_paged: page size must be 256")
    _req(k_in.dim() == 2 and v_in.dim() == 2 and k_in.stride(1) == 1 and v_in.stride(1) == 1 and k_in.dtype == torch.half and v_in.dtype == torch.half,
         "quant_cache_paged_strided: k_in / v_in must be 2-D fp16 views with unit column stride")
    dim = k_out.shape[-1] // _kv_bits(k_out, k_scales) * 32
    _req(k_in.shape[1] == dim and v_in.shape[1] == dim, "quant_cache_paged_strided: width must match the cache")
    bsz = block_table.shape[0]
    _req(k_in.shape[0] == bsz * seq_len and v_in.shape[0] == bsz * seq_len, "quant_cache_paged_strided: rows must be bsz * seq_len")
    _check(_lib.lib().exl3_quant_cache_paged_strided(_p(k_in), _p(k_out), _p(k_scales), _p(v_in), _p(v_out), _p(v_scales),
                                                     _p(cache_seqlens), _p(block_table), bsz, block_table.shape[1], page_size, seq_len, dim,
                                                     _kv_bits(k_out, k_scales), _kv_bits(v_out, v_scales), k_in.stride(0), v_in.stride(0), _stream(k_in)))


def rope_strided(q, k, inv_freq, position: int, positions, position_ids, attn_factor: float, bsz: int, seq_len: int):
    """In-place NEOX rope (head_dim 128, no head norm) on q (bsz * seq_len, heads_q * 128) and k (.., heads_k * 128), column ranges of a wider
    row-major fp16 matrix; rope.cu semantics otherwise."""
    _dev(q)
    _req(q.dim() == 2 and q.dtype == torch.half and q.stride(1) == 1 and q.shape[0] == bsz * seq_len and q.shape[1] % 128 == 0, "rope_strided: bad q")
    hk, ldk = 0, 0
    if k is not None:
        _req(k.dim() == 2 and k.dtype == torch.half and k.stride(1) == 1 and k.shape[0] == bsz * seq_len and k.shape[1] % 128 == 0, "rope_strided: bad k")
        hk, ldk = k.shape[1] // 128, k.stride(0)
    _req(inv_freq.numel() == 64 and inv_freq.dtype == torch.float, "rope_strided: inv_freq must be 64 floats")
    _check(_lib.lib().exl3_rope_strided(_p(q), _p(k), _p(inv_freq), bsz, seq_len, q.shape[1] // 128, hk, q.stride(0), ldk, int(position),
                                        _p(positions), _p(position_ids), float(attn_factor), _stream(q)))


def dequant_cache_paged(k_in, k_scales, k_out, v_in, v_scales, v_out, cache_seqlens, block_table, page_size: int,
                        sliding_window: int = 0, compand_a: float = 0.0):
    """cache/q_cache.cuh:79-92: rows [0, cache_seqlens[b]) of every sequence; sliding_window > 0 leaves the rows the reference skips untouched."""
    _dev(k_in)
    _req(page_size == 256, "dequant_cache_paged: page size must be 256")
    dim = k_in.shape[-1] // _kv_bits(k_in, k_scales) * 32
    bsz = block_table.shape[0]
    _check(_lib.lib().exl3_dequant_cache_paged_ex(_p(k_in), _p(k_scales), _p(k_out), _p(v_in), _p(v_scales), _p(v_out),
                                                  _p(cache_seqlens), _p(block_table), bsz, block_table.shape[1], page_size, dim,
                                                  _kv_bits(k_in, k_scales), _kv_bits(v_in, v_scales), int(sliding_window), float(compand_a), 0, 0,
                                                  _stream(k_in)))


def dequant_cache_paged_window(k_in, k_scales, k_out, v_in, v_scales, v_out, cache_seqlens, block_table, page_size: int,
                               bonus_len: int, compand_a: float = 0.0):
    """cache/q_cache.cuh:64-77: page p of sequence b is expanded densely into row (b * pages_per_seq + p) * page_size of the scratch k_out / v_out
    (at least bsz * pages_per_seq pages), rows up to cache_seqlens[b] + bonus_len."""
    _dev(k_in)
    _req(page_size == 256, "dequant_cache_paged_window: page size must be 256")
    dim = k_in.shape[-1] // _kv_bits(k_in, k_scales) * 32
    bsz, pps = block_table.shape
    _req(k_out.shape[0] >= bsz * pps and v_out.shape[0] >= bsz * pps, "scratch too small for block table span")
    _check(_lib.lib().exl3_dequant_cache_paged_ex(_p(k_in), _p(k_scales), _p(k_out), _p(v_in), _p(v_scales), _p(v_out),
                                                  _p(cache_seqlens), _p(block_table), bsz, pps, page_size, dim,
                                                  _kv_bits(k_in, k_scales), _kv_bits(v_in, v_scales), 0, float(compand_a), 1, int(bonus_len),
                                                  _stream(k_in)))


def silu_mul(g, u, y):
    """activation.cu silu_mul: y = silu(g) * u"""
    _dev(g)
    _req(g.shape == u.shape and g.dtype == u.dtype and g.dtype in (torch.half, torch.float), "silu_mul: bad inputs")
    _req(y.dtype == torch.half and y.numel() == g.numel(), "silu_mul: y must be float16")
    _check(_lib.lib().exl3_silu_mul(_p(g), _p(u), _p(y), g.numel(), int(g.dtype == torch.float), _stream(g)))


ACT_SILU, ACT_GELU, ACT_RELU2, ACT_RELU, ACT_SILU_OAI = 0, 1, 2, 3, 4


def act_mul(g, u, y, act: int, act_limit: float = 0.0):
    """activation.cu silu_mul / gelu_mul / relu2_mul / silu_oai_mul (x, y, z, act_limit): y = fp16(act(g) * u) with the reference's clamps."""
    _dev(g)
    _req(g.shape == u.shape and g.dtype == u.dtype and g.dtype in (torch.half, torch.float), "act_mul: bad inputs")
    _req(y.dtype == torch.half and y.numel() == g.numel(), "act_mul: y must be float16")
    _req(g.is_contiguous() and u.is_contiguous() and y.is_contiguous(), "act_mul: tensors must be contiguous")
    _check(_lib.lib().exl3_act_mul(_p(g), _p(u), _p(y), g.numel(), int(g.dtype == torch.float), int(act), float(act_limit), _stream(g)))


def gelu_mul(x, y, z, act_limit: float = 0.0):
    """activation.cu:246-254"""
    act_mul(x, y, z, ACT_GELU, act_limit)


def relu2_mul(x, y, z, act_limit: float = 0.0):
    """activation.cu:324-332"""
    act_mul(x, y, z, ACT_RELU2, act_limit)


def relu_mul(x, y, z, act_limit: float = 0.0):
    """activation.cu:336-378: z = relu(x) * y (x = y gives the non-gated relu^2 of NemotronH-style experts)"""
    act_mul(x, y, z, ACT_RELU, act_limit)


def silu_oai_mul(x, y, z, act_limit: float = 0.0):
    """activation.cu:168-176 (gpt-oss clamped swiglu)"""
    act_mul(x, y, z, ACT_SILU_OAI, act_limit)


def mul_sigmoid_(x, y):
    """activation.cu:526-568: x *= sigmoid(y), fp16, in place"""
    _dev(x)
    _req(x.dtype == torch.half and y.dtype == torch.half, "mul_sigmoid_: tensors must be float16")
    _req(x.numel() == y.numel() and x.is_contiguous() and y.is_contiguous() and x.numel() % 4 == 0, "mul_sigmoid_: contiguous tensors of equal size (a multiple of 4)")
    _check(_lib.lib().exl3_mul_gate(_p(x), _p(y), x.numel(), 0, 0, _stream(x)))


def _mul_gate_broadcast(x, y, softplus: int, name: str):
    _dev(x)
    _req(x.dtype == torch.half and y.dtype == torch.half, f"{name}: tensors must be float16")
    _req(x.dim() == 4, "x must be [B, S, H, D]")
    _req(y.dim() == 3, "y must be [B, S, H]")
    _req(x.shape[:3] == y.shape, "x and y have incompatible shapes")
    _req(x.is_contiguous() and y.is_contiguous(), "x and y must be contiguous")
    _req(x.shape[3] % 4 == 0, f"{name}: head_dim must be divisible by 4")
    _check(_lib.lib().exl3_mul_gate(_p(x), _p(y), x.numel(), x.shape[3], softplus, _stream(x)))


def mul_sigmoid_broadcast_(x, y):
    """activation.cu:572-614: x [B, S, H, D] *= sigmoid(y [B, S, H]) broadcast over D (headwise attention gate)"""
    _mul_gate_broadcast(x, y, 0, "mul_sigmoid_broadcast_")


def mul_softplus_broadcast_(x, y):
    """activation.cu:618-660: x [B, S, H, D] *= softplus(y [B, S, H]) broadcast over D; the gate is evaluated in fp32"""
    _mul_gate_broadcast(x, y, 1, "mul_softplus_broadcast_")


def add_sigmoid_gate(x, y, z):
    """activation.cu:480-524: z += x * sigmoid(y), fp32, one gate per row of x (y.size(-1) == 1)"""
    _dev(x)
    _req(x.dtype == torch.float and y.dtype == torch.float and z.dtype == torch.float, "add_sigmoid_gate: tensors must be float32")
    _req(y.shape[-1] == 1, "gate must have size(-1) == 1")
    _req(x.numel() == z.numel() and x.numel() == y.numel() * x.shape[-1] and x.is_contiguous() and y.is_contiguous() and z.is_contiguous(), "add_sigmoid_gate: bad shapes")
    _check(_lib.lib().exl3_add_sigmoid_gate(_p(x), _p(y), _p(z), x.numel(), x.shape[-1], _stream(x)))


def add_sigmoid_gate_proj(x, y, z, w):
    """activation.cu:662-714: z += x * sigmoid(y @ w): x, z fp32 (bsz, dim); y fp16 (bsz, dim); w fp16 (dim, 1)"""
    _dev(x)
    _req(x.dtype == torch.float and z.dtype == torch.float and y.dtype == torch.half and w.dtype == torch.half, "add_sigmoid_gate_proj: x, z float32; y, w float16")
    dim = x.shape[-1]
    _req(w.numel() == dim and y.numel() == x.numel() == z.numel() and x.is_contiguous() and y.is_contiguous() and z.is_contiguous() and w.is_contiguous(),
         "add_sigmoid_gate_proj: bad shapes")
    _check(_lib.lib().exl3_add_sigmoid_gate_proj(_p(x), _p(y), _p(z), _p(w), x.numel() // dim, dim, _stream(x)))


def paged_kv_cache_update(k, v, k_cache, v_cache, block_table, cache_seqlens):
    """generator/cache.cu:186-240: append k / v [B, S_new, H_k, D] fp16 to the fp16 paged cache [num_blocks, 256, H_k, D] at cache_seqlens[b] + t"""
    _dev(k)
    _req(k.dtype == torch.half and v.dtype == torch.half and k_cache.dtype == torch.half and v_cache.dtype == torch.half, "paged_kv_cache_update: tensors must be float16")
    _req(block_table.dtype == torch.int32 and cache_seqlens.dtype == torch.int32, "paged_kv_cache_update: block_table / cache_seqlens must be int32")
    _req(k.dim() == 4 and v.shape == k.shape, "k, v must have shape [B, S_new, H_k, D]")
    _req(k_cache.dim() == 4 and v_cache.shape == k_cache.shape and k_cache.shape[2:] == k.shape[2:], "k_cache / v_cache must have shape [num_blocks, 256, H_k, D]")
    _req(k_cache.shape[1] == 256, "this kernel needs page_size == 256")
    _req(block_table.dim() == 2 and cache_seqlens.dim() == 1 and block_table.shape[0] == k.shape[0] == cache_seqlens.shape[0], "block_table [B, max_blocks_per_seq], cache_seqlens [B]")
    _req(k.shape[3] % 8 == 0, "dim must be divisible by 8")
    _req(k.is_contiguous() and v.is_contiguous() and k_cache.is_contiguous() and v_cache.is_contiguous() and block_table.is_contiguous(), "paged_kv_cache_update: contiguous tensors")
    B, S, H, D = k.shape
    _check(_lib.lib().exl3_paged_kv_cache_update(_p(k), _p(v), _p(k_cache), _p(v_cache), _p(block_table), _p(cache_seqlens), B, S, H, D, block_table.shape[1], _stream(k)))


def deinterleave_qg(qg, q, g, head_dim: int):
    """activation.cu:716-785: [.., heads, (q: head_dim, g: head_dim)] -> contiguous q and g"""
    _dev(qg)
    _req(qg.dtype == torch.half and q.dtype == torch.half and g.dtype == torch.half, "deinterleave_qg: tensors must be float16")
    _req(qg.is_contiguous() and q.is_contiguous() and g.is_contiguous(), "deinterleave_qg: tensors must be contiguous")
    _req(head_dim % 8 == 0 and q.numel() == g.numel() and qg.numel() == 2 * q.numel() and q.numel() % head_dim == 0, "deinterleave_qg: incompatible shapes")
    _check(_lib.lib().exl3_deinterleave_qg(_p(qg), _p(q), _p(g), q.numel() // head_dim, int(head_dim), _stream(qg)))


def silu_mul_2d(g, u, y):
    """y = silu(g) * u for fp16 2-D views with unit column stride (g, u may be column ranges of one wider matrix)."""
    _dev(g)
    _req(g.dtype == torch.half and u.dtype == torch.half and y.dtype == torch.half, "silu_mul_2d: tensors must be float16")
    _req(g.dim() == 2 and g.shape == u.shape == y.shape and g.stride(1) == 1 and u.stride(1) == 1 and y.is_contiguous(), "silu_mul_2d: bad shapes / strides")
    _check(_lib.lib().exl3_silu_mul_2d(_p(g), _p(u), _p(y), g.shape[0], g.shape[1], g.stride(0), u.stride(0), _stream(g)))


def add(x, y):
    """add.cu: x += y"""
    _dev(x)
    _req(x.numel() == y.numel(), "add: size mismatch")
    _check(_lib.lib().exl3_add(_p(x), _p(y), x.numel(), int(x.dtype == torch.float), int(y.dtype == torch.float), _stream(x)))


def softcap(x, y, scale: float):
    """softcap.cu:59-100 (same argument order): y = scale * tanh(x / scale); fp16 or fp32, in place when y is x."""
    _dev(x)
    _req(x.dtype == y.dtype and x.dtype in (torch.half, torch.float), "softcap: x, y must both be float16 or float32")
    _req(x.numel() == y.numel() and x.is_contiguous() and y.is_contiguous(), "softcap: contiguous tensors of equal size")
    _check(_lib.lib().exl3_softcap(_p(x), _p(y), x.numel(), float(scale), int(x.dtype == torch.float), _stream(x)))


# --------------------------------------------------------------------------------------------------
# fused decode pipeline (deferred-epilogue GEMVs + glue kernels); see include/exl3_hip.h
# --------------------------------------------------------------------------------------------------

GEMV_IN_ROTATED, GEMV_OUT_DEFERRED, GEMV_IN_NORM = 1, 2, 4
GEMV_OUT_ATOMIC = 32      # (m <= 4) Cs[i] = int64 fixed-point accumulator (value * 2^32) the finished rows are added into: the "fx" decode pipeline


def _parr(ptrs):
    return (_vp * len(ptrs))(*[(p if isinstance(p, int) or p is None else p.data_ptr()) for p in ptrs])


def exl3_gemv_ex(A, xhs, xsums, Bs, Cs, suhs, svhs, m: int, mcg: bool, mul1: bool, flags: int, force_split: int = 0, c_fp32: bool = False):
    """Returns (slab_device_pointers, S).  A or xhs[i] is the input; Cs/svhs may be None with GEMV_OUT_DEFERRED."""
    ref = A if A is not None else xhs[0]
    _dev(ref)
    cnt = len(Bs)
    k, K = Bs[0].shape[0] * 16, Bs[0].shape[2] // 16
    _req(ref.shape[-1] == k and ref.numel() >= m * k, "exl3_gemv_ex: input width must match B (k)")
    _req(all(B.shape[0] * 16 == k and B.shape[2] // 16 == K for B in Bs), "exl3_gemv_ex: the matrices of one launch share k and bits per weight")
    ns = (ctypes.c_int * cnt)(*[B.shape[1] * 16 for B in Bs])
    slabs = (_vp * cnt)()
    S = ctypes.c_int(0)
    _check(_lib.lib().exl3_gemv_ex(_p(A), _parr(xhs) if xhs else None, _parr(xsums) if xsums else None, _parr(Bs),
                                   _parr(Cs) if Cs else None, _parr(suhs) if suhs else None, _parr(svhs) if svhs else None, None,
                                   ns, cnt, m, k, K, _cb(mcg, mul1), int(c_fp32), flags, force_split, slabs, ctypes.byref(S), _stream(ref)))
    return [int(s) if s else 0 for s in slabs], S.value


def glue_norm(y_slab: int | None, y_S: int, svh, bias, resid, w, eps: float, suhs, xhs, xsums, m: int, xn_out=None, y_dense=None):
    _dev(resid)
    hidden = resid.shape[-1]
    _check(_lib.lib().exl3_glue_norm(y_slab, y_S, _p(y_dense), _p(svh), _p(bias), _p(resid), _p(w), float(eps), _parr(suhs), _parr(xhs), _parr(xsums),
                                     len(suhs), m, hidden, _p(xn_out), _stream(resid)))


def glue_qkv(slabs, S: int, svh_q, svh_k, svh_v, q_out, k_out, v_out, inv_freq, positions, k_cache, k_scales, v_cache, v_scales,
             block_table, page_size: int, k_bits: int, v_bits: int, m: int, heads_q: int, heads_kv: int, head_dim: int,
             rope_mode: int = 2, attn_factor: float = 1.0, tab=None):
    if tab is not None:
        return glue_qkv_rs(slabs, S, svh_q, svh_k, svh_v, q_out, k_out, v_out, inv_freq, positions, k_cache, k_scales, v_cache, v_scales,
                           block_table, page_size, k_bits, v_bits, m, heads_q, heads_kv, head_dim, None, None, 0, 0.0, rope_mode, attn_factor, tab)
    _dev(q_out)
    _check(_lib.lib().exl3_glue_qkv(slabs[0], slabs[1], slabs[2], S, _p(svh_q), _p(svh_k), _p(svh_v), _p(q_out), _p(k_out), _p(v_out),
                                    _p(inv_freq), _p(positions), _p(k_cache), _p(k_scales), _p(v_cache), _p(v_scales), _p(block_table),
                                    block_table.shape[1] if block_table is not None else 0, page_size, k_bits, v_bits, m, heads_q, heads_kv,
                                    head_dim, rope_mode, float(attn_factor), _stream(q_out)))


def glue_qkv_rs(slabs, S: int, svh_q, svh_k, svh_v, q_out, k_out, v_out, inv_freq, positions, k_cache, k_scales, v_cache, v_scales,
                block_table, page_size: int, k_bits: int, v_bits: int, m: int, heads_q: int, heads_kv: int, head_dim: int,
                ss_prev, ss_new, hidden: int, eps: float, rope_mode: int = 2, attn_factor: float = 1.0, tab=None):
    """glue_qkv for slabs written by exl3_gemv_ex_resid: q, k, v are multiplied by rsqrt(ms_new + eps) / rsqrt(ms_prev + eps) of their row.
    tab = (sin, cos, slots) of qkv_prep: the kernel reads the per-step tables instead of computing sin / cos and the cache row itself."""
    _dev(q_out)
    sn, cs, sl = tab if tab is not None else (None, None, None)
    _check(_lib.lib().exl3_glue_qkv_tab(slabs[0], slabs[1], slabs[2], S, _p(svh_q), _p(svh_k), _p(svh_v), _p(q_out), _p(k_out), _p(v_out),
                                        _p(inv_freq), _p(positions), _p(k_cache), _p(k_scales), _p(v_cache), _p(v_scales), _p(block_table),
                                        block_table.shape[1] if block_table is not None else 0, page_size, k_bits, v_bits, m, heads_q, heads_kv,
                                        head_dim, rope_mode, float(attn_factor), _p(ss_prev), _p(ss_new), int(hidden), float(eps),
                                        _p(sn), _p(cs), _p(sl), _stream(q_out)))


def qkv_prep(inv_freq, positions, head_dim: int, block_table, page_size: int, sin_out, cos_out, slots, attn_factor: float = 1.0):
    """Per-step tables for glue_qkv(tab=...): sin / cos (m, 64) fp32 and the cache row (int64 (m,)) of every token of the step."""
    _dev(sin_out)
    _req(sin_out.dtype == torch.float and cos_out.dtype == torch.float and sin_out.shape == cos_out.shape and sin_out.shape[-1] == 64, "qkv_prep: sin / cos must be (m, 64) float32")
    _req(slots is None or (slots.dtype == torch.long and block_table is not None and block_table.dtype == torch.int32), "qkv_prep: slots must be int64 and need an int32 block table")
    _check(_lib.lib().exl3_qkv_prep(_p(inv_freq), _p(positions), float(attn_factor), positions.shape[0], int(head_dim), _p(block_table),
                                    block_table.shape[1] if block_table is not None else 0, int(page_size), _p(sin_out), _p(cos_out), _p(slots), _stream(sin_out)))


def glue_act(slabs, S: int, svh_g, svh_u, suh_d, xh_d, xsum_d, m: int, a_out=None):
    _dev(xh_d)
    _check(_lib.lib().exl3_glue_act(slabs[0], slabs[1], S, _p(svh_g), _p(svh_u), _p(suh_d), _p(xh_d), _p(xsum_d), _p(a_out), m,
                                    xh_d.shape[-1], _stream(xh_d)))


def glue_act_rs(slabs, S: int, svh_g, svh_u, suh_d, xh_d, xsum_d, m: int, ss_prev, ss_new, hidden: int, eps: float, a_out=None):
    """glue_act for gate / up slabs of a row that glue_resid_rotate normalised with the previous residual's 1/rms: g, u are multiplied by
    rsqrt(ms_new + eps) / rsqrt(ms_prev + eps) of their row first."""
    _dev(xh_d)
    _check(_lib.lib().exl3_glue_act_rs(slabs[0], slabs[1], S, _p(svh_g), _p(svh_u), _p(suh_d), _p(xh_d), _p(xsum_d), _p(a_out), m,
                                       xh_d.shape[-1], _p(ss_prev), _p(ss_new), int(hidden), float(eps), _stream(xh_d)))


def glue_resid_rotate(y_slab, y_S: int, svh, bias, resid, ss_prev, ss_new, w, eps: float, suhs, xhs, m: int, xsums=None, y_dense=None):
    """glue_resid + glue_rotate in one launch (batches above 4 rows): resid += y, ss_new = block sums of squares of the new residual,
    xh_i = had128(fp16(resid_new * w * r_prev) * suh_i) with r_prev from ss_prev (another buffer than ss_new); finish the consumers with
    glue_qkv_rs / glue_act_rs (ss_prev, ss_new)."""
    _dev(resid)
    _check(_lib.lib().exl3_glue_resid_rotate(y_slab, y_S, _p(y_dense), _p(svh), _p(bias), _p(resid), _p(ss_prev), _p(ss_new), _p(w), float(eps),
                                             _parr(suhs), _parr(xhs), _parr(xsums) if xsums else None, len(suhs), m, resid.shape[-1], _stream(resid)))


# ---- GEMV launches with an in-kernel tail epilogue (one launch per sublayer boundary of a decode step) ----------------------

def _kK(B):
    return B.shape[0] * 16, B.shape[2] // 16


def rope_table(inv_freq, positions, sin_out, cos_out, attn_factor: float = 1.0):
    _dev(sin_out)
    _check(_lib.lib().exl3_rope_table(_p(inv_freq), _p(positions), float(attn_factor), positions.shape[0], _p(sin_out), _p(cos_out), _stream(sin_out)))


def exl3_gemv_norm(A, xh, xsum, B, suh, svh, bias, m: int, mcg: bool, mul1: bool, resid, norm_w, eps: float, t_suhs, t_xhs, t_xsums, xn_out=None):
    """o_proj / down_proj + residual add + RMSNorm + input Hadamard of the next linears, one launch."""
    _dev(resid)
    k, K = _kK(B)
    _check(_lib.lib().exl3_gemv_norm(_p(A), _p(xh), _p(xsum), _p(B), _p(suh), _p(svh), _p(bias), m, k, B.shape[1] * 16, K, _cb(mcg, mul1),
                                     _p(resid), _p(norm_w), float(eps), _parr(t_suhs) if t_suhs else None, _parr(t_xhs) if t_xhs else None,
                                     _parr(t_xsums) if t_xsums else None, len(t_suhs) if t_suhs else 0, _p(xn_out), _stream(resid)))


def exl3_gemv_act(A, xhs, xsums, Bs, suhs, svhs, m: int, mcg: bool, mul1: bool, suh_d, xh_d, xsum_d, a_out=None):
    """gate_proj + up_proj + silu(g) * u + input Hadamard of down_proj, one launch."""
    _dev(xh_d)
    k, K = _kK(Bs[0])
    _check(_lib.lib().exl3_gemv_act(_p(A), _parr(xhs) if xhs else None, _parr(xsums) if xsums else None, _parr(Bs), _parr(suhs) if suhs else None,
                                    _parr(svhs), m, k, Bs[0].shape[1] * 16, K, _cb(mcg, mul1), _p(suh_d), _p(xh_d), _p(xsum_d), _p(a_out),
                                    _stream(xh_d)))


def exl3_gemv_qkv(A, xhs, xsums, Bs, suhs, svhs, m: int, mcg: bool, mul1: bool, q_out, k_out, v_out, rope_sin, rope_cos, positions,
                  k_cache, k_scales, v_cache, v_scales, block_table, page_size: int, k_bits: int, v_bits: int,
                  heads_q: int, heads_kv: int, head_dim: int, rope_mode: int = 2):
    """q/k/v projections + RoPE + quantized KV-cache append, one launch."""
    _dev(q_out)
    k, K = _kK(Bs[0])
    _check(_lib.lib().exl3_gemv_qkv(_p(A), _parr(xhs) if xhs else None, _parr(xsums) if xsums else None, _parr(Bs), _parr(suhs) if suhs else None,
                                    _parr(svhs), m, k, K, _cb(mcg, mul1), _p(q_out), _p(k_out), _p(v_out), _p(rope_sin), _p(rope_cos), _p(positions),
                                    _p(k_cache), _p(k_scales), _p(v_cache), _p(v_scales), _p(block_table),
                                    block_table.shape[1] if block_table is not None else 0, page_size, k_bits, v_bits, heads_q, heads_kv,
                                    head_dim, rope_mode, _stream(q_out)))


def exl3_gemv_ex_norm(resid, norm_w, ss_part, eps: float, Bs, Cs, suhs, svhs, m: int, mcg: bool, mul1: bool, flags: int = 0,
                      force_split: int = 0, c_fp32: bool = False):
    """exl3_gemv_ex on rms_norm(resid) * norm_w, the norm applied inside the GEMV (ss_part from glue_resid).  Returns (slabs, S)."""
    _dev(resid)
    cnt = len(Bs)
    k, K = _kK(Bs[0])
    ns = (ctypes.c_int * cnt)(*[B.shape[1] * 16 for B in Bs])
    slabs = (_vp * cnt)()
    S = ctypes.c_int(0)
    _check(_lib.lib().exl3_gemv_ex_norm(_p(resid), _p(norm_w), _p(ss_part), float(eps), _parr(Bs), _parr(Cs) if Cs else None, _parr(suhs),
                                        _parr(svhs) if svhs else None, None, ns, cnt, m, k, K, _cb(mcg, mul1), int(c_fp32), flags, force_split,
                                        slabs, ctypes.byref(S), _stream(resid)))
    return [int(s) if s else 0 for s in slabs], S.value


def glue_rotate(resid, ss_part, w, eps: float, suhs, xhs, m: int, xsums=None):
    """xh_i = had128(rms_norm(resid) * w * suh_i) for up to 3 consumers (batches above 4 rows; see exl3_glue.hip G1b)."""
    _dev(resid)
    _check(_lib.lib().exl3_glue_rotate(_p(resid), _p(ss_part), _p(w), float(eps), _parr(suhs), _parr(xhs), _parr(xsums) if xsums else None,
                                       len(suhs), m, resid.shape[-1], _stream(resid)))


def glue_resid(y_slab, y_S: int, svh, bias, resid, ss_part, m: int, y_dense=None):
    _dev(resid)
    _check(_lib.lib().exl3_glue_resid(y_slab, y_S, _p(y_dense), _p(svh), _p(bias), _p(resid), _p(ss_part), m, resid.shape[-1], _stream(resid)))


# --------------------------------------------------------------------------------------------------
# BC_GatedMLP (libtorch/mlp.h:20-118, mlp.cpp:14-130): d = down(act(gate(x)) * up(x)) for 1..MAX_BSZN tokens
# --------------------------------------------------------------------------------------------------

MAX_BSZN = 8        # libtorch/mlp.h:11 (this build accepts up to 16 rows, the GEMV pass size)


class BC_GatedMLP:
    """Mirror of the reference's decode-time gated-MLP runner.  Constructor arguments follow libtorch/mlp.h:53-72.  The reference
    runs exl3_mgemm -> silu_mul -> exl3_gemm (+ bias adds) as separate graph nodes; here the SiLU path without biases is three
    launches with no intermediate round trip of g/u: gate/up GEMV with deferred epilogue -> glue_act (split-k reduce, output
    Hadamards, silu(g)*u, input Hadamard of down) -> down GEMV on the pre-rotated input.  `a` still receives silu(g)*u (the
    reference's observable intermediate).  GELU / relu2 / an act_limit take the reference's launch-per-op route (gate, up, act_mul, down)."""

    def __init__(self, guh, gu, a, down_xh, gu_ptrs_trellis, gu_ptrs_suh, gu_ptrs_svh, gu_K, gu_mcg, gu_mul1,
                 act_silu, act_gelu, act_relu2, gate, up, down, act_limit):
        _req(gu_ptrs_trellis is not None or (gate is not None and up is not None),
             "BC_GatedMLP: need fused mgemm tensors or gate/up handles")
        _req(gate is not None and up is not None and down is not None, "BC_GatedMLP: this build needs the gate/up/down handles")
        _req(int(bool(act_silu)) + int(bool(act_gelu)) + int(bool(act_relu2)) == 1, "BC_GatedMLP: exactly one of act_silu / act_gelu / act_relu2")
        self.act = ACT_SILU if act_silu else (ACT_GELU if act_gelu else ACT_RELU2)
        self.act_limit = float(act_limit)
        self.guh, self.gu, self.a, self.down_xh = guh, gu, a, down_xh
        self.gate, self.up, self.down = gate, up, down
        self._xs = None

    def run_bszN(self, x: torch.Tensor, d: torch.Tensor):
        _dev(x)
        hidden = x.shape[-1]
        m = x.numel() // hidden
        _req(1 <= m <= 16, "run_bszN: bsz out of supported range")
        g, u, dn = self.gate, self.up, self.down
        inter = g.trellis.shape[1] * 16
        x2 = x.view(m, hidden)
        a_n = self.a.view(-1, inter)[:m]
        xh_n = self.down_xh.view(-1, inter)[:m]
        _req(a_n.shape[0] == m and xh_n.shape[0] == m, "run_bszN: scratch buffers too small for this bsz")
        if g.bias is not None or u.bias is not None or g.K != u.K or g.mcg != u.mcg or g.mul1 != u.mul1 or self.act != ACT_SILU or self.act_limit != 0.0:
            # op-by-op route (reference's non-mgemm branch, mlp.cpp:62-77; also every activation other than the plain SiLU)
            gu_n = self.gu.view(2, -1, inter)[:, :m].contiguous() if self.gu.numel() >= 2 * m * inter else torch.empty((2, m, inter), dtype=torch.half, device=x.device)
            g.run(x2, gu_n[0]); u.run(x2, gu_n[1])
            act_mul(gu_n[0], gu_n[1], a_n, self.act, self.act_limit)
            dn.run(a_n, d.view(m, -1))
            return
        if self._xs is None or self._xs.shape[0] < m:
            self._xs = torch.empty((16, inter // 128), dtype=torch.float, device=x.device)
        xs = self._xs[:m]
        slabs, S = exl3_gemv_ex(x2, None, None, [g.trellis, u.trellis], None, [g.suh, u.suh], None, m, g.mcg, g.mul1, GEMV_OUT_DEFERRED)
        glue_act(slabs, S, g.svh, u.svh, dn.suh, xh_n, xs, m, a_out=a_n)
        d2 = d.view(m, -1)
        exl3_gemv_ex(None, [xh_n], [xs], [dn.trellis], [d2], None, [dn.svh], m, dn.mcg, dn.mul1, GEMV_IN_ROTATED, c_fp32=(d.dtype == torch.float))
        if dn.bias is not None:
            add(d2, dn.bias.view(1, -1).expand(m, -1).contiguous() if m > 1 else dn.bias)


class BC_BlockSparseMLP:
    """libtorch/blocksparse_mlp.h:24-205, blocksparse_mlp.cpp:66-696: the sparse-MoE block's runner the reference's modules/block_sparse_mlp.py:917
    constructs -- same constructor argument list (the statics of the Python side) and the same three entry points, each the reference's own op
    sequence over this module's ops:
      run_bszN(y, selected_experts, routing_weights)    1 .. MAX_BSZN tokens: indexed exl3_mgemm for gate and up over the selected experts (expert
                                                        range [min_expert, max_expert) under expert parallelism), activation, weighted indexed down
                                                        exl3_mgemm whose rows 0 .. tokens - 1 of out_d hold the reductions (+ shared experts)
      run_single_expert(y, e)                            one expert's MLP on <= 32 rows through the quantized kernels, result in out_d2[:rows]
      run_single_expert_dq(y, e, yh, interm, interm_a, out)   the same through reconstruct + hgemm (large row counts)
    Not covered (it raises at construction): per-expert biases (gpt-oss).  The reference
    captures one CUDA graph per batch size inside the class; here run_*() is plain launches -- capture the surrounding step in one hipGraph."""

    MAX_BSZN = 8                                                      # mlp.h:11 (must match the Python side's buffer sizes)
    TEMP_ROWS_GRAPH = 32

    def __init__(self, yh2, yh, interm_gu, interm_g, interm_u, interm_a, interm_a2, out_d, out_d2, out_d_sh, z, dq_temp_up, dq_temp_down,
                 min_expert, max_expert, gate_ptrs_trellis, gate_ptrs_suh, gate_ptrs_svh, gate_K, gate_mcg, gate_mul1,
                 up_ptrs_trellis, up_ptrs_suh, up_ptrs_svh, up_K, up_mcg, up_mul1,
                 down_ptrs_trellis, down_ptrs_suh, down_ptrs_svh, down_K, down_mcg, down_mul1,
                 act_silu, act_gelu, act_silu_oai, shared_experts, shared_gate, act_limit, gates, ups, downs,
                 gu_trellis_ptr, gu_suh_ptr, gu_svh_ptr, a_gather, gate_bias_ptrs=None, up_bias_ptrs=None, down_bias_ptrs=None,
                 y_pad=None, out_trim=None, act_relu2=False):
        self.gated = len(gates) > 0
        _req(self.gated or act_relu2, "BC_BlockSparseMLP: gateless experts require act_relu2")                                  # blocksparse_mlp.cpp:493
        _req(not (shared_experts is not None and (down_bias_ptrs is not None or y_pad is not None)),
             "BC_BlockSparseMLP: shared experts not supported with expert biases or padded dims")                               # :494
        _req(gate_bias_ptrs is None and up_bias_ptrs is None and down_bias_ptrs is None, "BC_BlockSparseMLP: per-expert biases are outside this build")
        _req(max_expert <= 512, "BC_BlockSparseMLP: Too many experts")
        self.yh2, self.yh, self.interm_gu, self.interm_g, self.interm_u, self.interm_a, self.interm_a2 = yh2, yh, interm_gu, interm_g, interm_u, interm_a, interm_a2
        self.out_d, self.out_d2, self.out_d_sh, self.dq_temp_up, self.dq_temp_down = out_d, out_d2, out_d_sh, dq_temp_up, dq_temp_down
        self.min_expert, self.max_expert = int(min_expert), int(max_expert)
        self.gate_t = (gate_ptrs_trellis, gate_ptrs_suh, gate_ptrs_svh, int(gate_K), bool(gate_mcg), bool(gate_mul1))
        self.up_t = (up_ptrs_trellis, up_ptrs_suh, up_ptrs_svh, int(up_K), bool(up_mcg), bool(up_mul1))
        self.down_t = (down_ptrs_trellis, down_ptrs_suh, down_ptrs_svh, int(down_K), bool(down_mcg), bool(down_mul1))
        self.act = (ACT_RELU if not self.gated else ACT_SILU if act_silu else ACT_GELU if act_gelu else ACT_SILU_OAI if act_silu_oai else ACT_RELU2)
        _req(not self.gated or act_silu or act_gelu or act_silu_oai or act_relu2, "BC_BlockSparseMLP: no activation selected")
        self.act_limit = float(act_limit)
        self.shared_experts, self.shared_gate, self.gates, self.ups, self.downs = shared_experts, shared_gate, list(gates), list(ups), list(downs)
        self.a_gather, self.y_pad, self.out_trim = a_gather, y_pad, out_trim
        self._flat_token = {}

    def _act(self, g, u, a):
        # non-gated: relu(u) * u = relu^2(u) (blocksparse_mlp.cpp:186-188); gated: the activation named at construction, with its clamps
        act_mul(u if not self.gated else g, u, a, self.act, self.act_limit)

    def run_bszN(self, y: torch.Tensor, selected_experts: torch.Tensor, routing_weights: torch.Tensor):
        _dev(y)
        num_tokens = int(y.shape[0])
        _req(1 <= num_tokens <= self.MAX_BSZN, "run_bszN: bsz out of supported range")
        numex = int(selected_experts.shape[-1])
        bszm = num_tokens * numex
        if num_tokens == 1:
            src = y
            if self.y_pad is not None:                                    # padded hidden dim: stage through the zero-padded static (blocksparse_mlp.cpp:86-95)
                self.y_pad[0:1, :y.shape[1]].copy_(y)
                src = self.y_pad[0:1]
            x_dense = src.unsqueeze(0)
            yi = src.view(1, 1, -1)
        else:
            ft = self._flat_token.get(num_tokens)
            if ft is None:
                ft = torch.arange(num_tokens, dtype=torch.long, device=y.device).unsqueeze(1).expand(num_tokens, numex).reshape(-1).contiguous()
                self._flat_token[num_tokens] = ft
            src = y
            if self.y_pad is not None:
                self.y_pad[:num_tokens, :y.shape[1]].copy_(y)
                src = self.y_pad[:num_tokens]
            x_dense = src.unsqueeze(0)
            ag = self.a_gather[:bszm]
            ag[:, :src.shape[1]].copy_(src.index_select(0, ft))         # every (token, slot) pair gets a copy of its token's row
            yi = ag.view(bszm, 1, ag.shape[1])
        yh_n, g_n, u_n, a_n, o_n = self.yh[:bszm], self.interm_g[:bszm], self.interm_u[:bszm], self.interm_a[:bszm], self.out_d[:bszm]
        sel, w = selected_experts.reshape(1, -1), routing_weights.reshape(1, -1)
        if self.gated:
            pt, ps, pv, K, mcg, mul1 = self.gate_t
            exl3_mgemm(yi, pt, g_n, ps, yh_n, pv, sel, None, K, -1, mcg, mul1, self.min_expert, self.max_expert, 0, num_tokens)
        pt, ps, pv, K, mcg, mul1 = self.up_t
        exl3_mgemm(yi, pt, u_n, ps, yh_n, pv, sel, None, K, -1, mcg, mul1, self.min_expert, self.max_expert, 0, num_tokens)
        self._act(g_n, u_n, a_n)
        pt, ps, pv, K, mcg, mul1 = self.down_t
        exl3_mgemm(a_n, pt, o_n, ps, g_n, pv, sel, w, K, -1, mcg, mul1, self.min_expert, self.max_expert, 0, num_tokens)
        if self.out_trim is not None:
            self.out_trim[:num_tokens].copy_(o_n[:num_tokens].squeeze(1)[:, :self.out_trim.shape[1]])
        if self.shared_experts is not None:
            sh = self.out_d_sh[:, :num_tokens]
            self.shared_experts.run_bszN(x_dense, sh)
            if self.shared_gate is not None:                              # z += shared(x) * sigmoid(x . w_gate)  (blocksparse_mlp.cpp:238-241)
                add_sigmoid_gate_proj(sh.view(num_tokens, -1), x_dense.view(num_tokens, -1), o_n[:num_tokens].view(num_tokens, -1), self.shared_gate.weight)
            else:
                add(o_n[:num_tokens].view(num_tokens, -1), sh.view(num_tokens, -1))

    def run_single_expert(self, y: torch.Tensor, expert_idx: int):
        _dev(y)
        bsz = int(y.shape[0])
        _req(bsz <= self.TEMP_ROWS_GRAPH, "run_single_expert: too many rows")
        ai, oi = self.interm_a2[:bsz], self.out_d2[:bsz]
        gi, ui = self.interm_gu[:bsz], self.interm_gu[bsz:2 * bsz]
        e = int(expert_idx)
        if self.gated:
            self.gates[e].run(y, gi)
        self.ups[e].run(y, ui)
        self._act(gi, ui, ai)
        self.downs[e].run(ai, oi)

    def run_single_expert_dq(self, y: torch.Tensor, expert_idx: int, yh: torch.Tensor, interm: torch.Tensor, interm_a: torch.Tensor, out: torch.Tensor):
        """blocksparse_mlp.cpp:640-696: rotate -> reconstruct W_hat into the shared scratch -> hgemm -> rotate back, per matrix."""
        _dev(y)
        bsz, e = int(y.shape[0]), int(expert_idx)
        yh1, yh2 = yh[:bsz], yh[bsz:2 * bsz]
        i1, i2 = interm[:bsz], interm[bsz:2 * bsz]
        up, dn = self.ups[e], self.downs[e]
        if self.gated:
            gt = self.gates[e]
            had_r_128(y, yh1, gt.suh, None, 1.0); had_r_128(y, yh2, up.suh, None, 1.0)
            reconstruct(self.dq_temp_up, gt.trellis, self.gate_t[3], self.gate_t[4], self.gate_t[5]); hgemm(yh1, self.dq_temp_up, i1)
            reconstruct(self.dq_temp_up, up.trellis, self.up_t[3], self.up_t[4], self.up_t[5]); hgemm(yh2, self.dq_temp_up, i2)
            had_r_128(i1, i1, None, gt.svh, 1.0); had_r_128(i2, i2, None, up.svh, 1.0)
        else:
            had_r_128(y, yh2, up.suh, None, 1.0)
            reconstruct(self.dq_temp_up, up.trellis, self.up_t[3], self.up_t[4], self.up_t[5]); hgemm(yh2, self.dq_temp_up, i2)
            had_r_128(i2, i2, None, up.svh, 1.0)
        self._act(i1, i2, interm_a)
        had_r_128(interm_a, interm_a, dn.suh, None, 1.0)
        reconstruct(self.dq_temp_down, dn.trellis, self.down_t[3], self.down_t[4], self.down_t[5]); hgemm(interm_a, self.dq_temp_down, out)
        had_r_128(out, out, None, dn.svh, 1.0)


class BC_MLP:
    """libtorch/mlp.h:116-190, mlp.cpp:149-186: the non-gated MLP's bsz-1 runner -- up -> activation (in place, against a ones tensor) -> down, with the
    zero-padded input / padded output statics of models whose widths are not multiples of 128.  Same keyword arguments as the reference's class."""

    def __init__(self, xp, u, ones, yp, act_silu, act_gelu, act_relu2, up, down, act_limit, hidden_size, out_size):
        _req(int(bool(act_silu)) + int(bool(act_gelu)) + int(bool(act_relu2)) == 1, "BC_MLP: exactly one of act_silu / act_gelu / act_relu2")
        self.xp, self.u, self.ones, self.yp, self.up, self.down = xp, u, ones, yp, up, down
        self.act = ACT_SILU if act_silu else (ACT_GELU if act_gelu else ACT_RELU2)
        self.act_limit, self.hidden_size, self.out_size = float(act_limit), int(hidden_size), int(out_size)

    def run_bsz1(self, x: torch.Tensor, d: torch.Tensor):
        _dev(x)
        x_in = x
        if self.xp is not None:
            self.xp[:, :self.hidden_size].copy_(x.view(1, self.hidden_size))
            x_in = self.xp
        u2 = self.u.view(1, -1)
        self.up.run(x_in.view(1, -1), u2)
        act_mul(u2, self.ones.view(1, -1), u2, self.act, self.act_limit)
        if self.yp is not None:
            self.down.run(u2, self.yp.view(1, -1))
            d.view(1, self.out_size).copy_(self.yp.view(1, -1)[:, :self.out_size])
        else:
            self.down.run(u2, d.view(1, -1))


def routing_std(hidden, gate, scores, topk_indices, topk_weights, per_expert_scale=None, gate_t=None, bias=None, gu_slots=None):
    """routing.cu:955-1010 (same argument order): scores = hidden @ gate, top-K by logit, softmax over the selected logits.
    gu_slots (this build, optional): int64 [2][bsz * K] receiving [selected | selected + experts], the slot list of one indexed exl3_mgemm over
    concatenated gate | up pointer tables."""
    _dev(hidden)
    _req(per_expert_scale is None or (per_expert_scale.dtype == torch.bfloat16 and per_expert_scale.numel() == scores.shape[1] and per_expert_scale.is_contiguous()),
         "routing_std: per_expert_scale must be a contiguous bfloat16 tensor with one entry per expert")
    _req(hidden.dtype == torch.half and gate.dtype == torch.half and scores.dtype == torch.half, "routing_std: hidden, gate, scores must be float16")
    _req(topk_indices.dtype == torch.long and topk_weights.dtype == torch.half, "routing_std: topk_indices int64, topk_weights float16")
    _req(gate.dim() == 2 and gate.shape[0] == hidden.shape[-1] and gate.shape[1] == scores.shape[-1], "routing_std: gate must be (hidden, experts)")
    _req(topk_indices.shape == topk_weights.shape and scores.shape[0] == topk_indices.shape[0], "routing_std: shape mismatch")
    _req(hidden.is_contiguous() and gate.is_contiguous() and scores.is_contiguous(), "routing_std: tensors must be contiguous")
    bsz = scores.shape[0]
    if gu_slots is not None:
        _req(gu_slots.dtype == torch.long and gu_slots.is_contiguous() and gu_slots.numel() == 2 * topk_indices.numel(), "routing_std: gu_slots must be int64 [2][bsz * K]")
    _check(_lib.lib().exl3_routing_std_scaled(_p(hidden), _p(gate), _p(bias), _p(per_expert_scale), _p(scores), _p(topk_indices), _p(topk_weights),
                                              _p(gu_slots), bsz, hidden.shape[-1], scores.shape[1], topk_indices.shape[1], _stream(hidden)))


def exl3_mgemm_act_deferred(G, U, B, suh, indices, K: int, mcg: int, mul1: int, n: int):
    """exl3_mgemm_act whose epilogue is deferred: returns (slab address, S) of the raw split-k slabs [bszm][n/128][S][m][128] fp32 for
    glue_resid_moe (svh, routing weights and the slot sum are applied there)."""
    _dev(G)
    _req(G.dtype == torch.half and U.dtype == torch.half and G.shape == U.shape and G.dim() == 3 and G.is_contiguous() and U.is_contiguous(),
         "exl3_mgemm_act_deferred: G, U must be contiguous float16 [bszm, m, k]")
    _req(B.dtype == torch.long and suh.dtype == torch.long and indices.dtype == torch.long and indices.is_contiguous(), "exl3_mgemm_act_deferred: int64 tables")
    bszm, m, k = G.shape
    _req(indices.numel() >= bszm, "exl3_mgemm_act_deferred: one index per slot")
    slab = (_vp * 1)()
    S = ctypes.c_int(0)
    _check(_lib.lib().exl3_mgemm_indexed_act_deferred(_p(G), _p(U), _p(B), _p(suh), _p(indices), bszm, m, k, int(n), int(K), _cb(bool(mcg), bool(mul1)),
                                                      slab, ctypes.byref(S), _stream(G)))
    return int(slab[0]), S.value


def glue_resid_moe(slab: int, S: int, svh_table, indices, weights, resid, ss_part, tokens: int):
    """Finishes a deferred weighted down launch of a MoE block and adds it to the residual stream (see exl3_glue_resid_moe)."""
    _dev(resid)
    _req(svh_table.dtype == torch.long and indices.dtype == torch.long and weights.dtype == torch.half and resid.dtype == torch.half, "glue_resid_moe: bad dtypes")
    _req(indices.is_contiguous() and weights.is_contiguous() and indices.numel() == weights.numel() and indices.numel() % tokens == 0, "glue_resid_moe: bad routing tensors")
    _check(_lib.lib().exl3_glue_resid_moe(slab, S, _p(svh_table), _p(indices), _p(weights), indices.numel() // tokens, _p(resid), _p(ss_part), tokens,
                                          resid.shape[-1], _stream(resid)))


def routing_std_norm(resid, norm_w, ss_part, eps: float, xn_out, gate, scores, topk_indices, topk_weights, bias=None, gu_slots=None):
    """routing_std on rms_norm(resid) formed inside the launch (mean square from ss_part of glue_resid); xn_out receives the normalised rows."""
    _dev(resid)
    _req(resid.dtype == torch.half and norm_w.dtype == torch.half and xn_out.dtype == torch.half and gate.dtype == torch.half and scores.dtype == torch.half,
         "routing_std_norm: resid, norm_w, xn_out, gate, scores must be float16")
    _req(ss_part.dtype == torch.float and topk_indices.dtype == torch.long and topk_weights.dtype == torch.half, "routing_std_norm: bad dtypes")
    _req(resid.is_contiguous() and xn_out.is_contiguous() and gate.is_contiguous() and scores.is_contiguous() and xn_out.shape == resid.shape, "routing_std_norm: bad layout")
    bsz = scores.shape[0]
    _check(_lib.lib().exl3_routing_std_norm(_p(resid), _p(norm_w), _p(ss_part), float(eps), _p(xn_out), _p(gate), _p(bias), _p(scores), _p(topk_indices),
                                            _p(topk_weights), _p(gu_slots), bsz, resid.shape[-1], scores.shape[1], topk_indices.shape[1], _stream(resid)))


def routing_std_fx(resid_fx, norm_w, ss_out, eps: float, xn_out, gate, scores, topk_indices, topk_weights, bias=None, gu_slots=None):
    """routing_std_norm of the fx pipeline: resid_fx = the int64 fixed-point residual accumulator [bsz, hidden]; the launch takes the row's exact mean
    square itself, leaves rms_norm(residual) in xn_out and the block sums of squares in ss_out [bsz, hidden / 128]."""
    _dev(resid_fx)
    _req(resid_fx.dtype == torch.long and norm_w.dtype == torch.half and xn_out.dtype == torch.half and gate.dtype == torch.half and scores.dtype == torch.half,
         "routing_std_fx: resid_fx int64; norm_w, xn_out, gate, scores float16")
    _req(ss_out.dtype == torch.float and topk_indices.dtype == torch.long and topk_weights.dtype == torch.half, "routing_std_fx: bad dtypes")
    _req(resid_fx.is_contiguous() and xn_out.is_contiguous() and gate.is_contiguous() and scores.is_contiguous() and xn_out.shape == resid_fx.shape
         and ss_out.is_contiguous() and ss_out.numel() >= resid_fx.numel() // 128, "routing_std_fx: bad layout")
    bsz = scores.shape[0]
    _check(_lib.lib().exl3_routing_std_fx(_p(resid_fx), _p(norm_w), _p(ss_out), float(eps), _p(xn_out), _p(gate), _p(bias), _p(scores), _p(topk_indices),
                                          _p(topk_weights), _p(gu_slots), bsz, resid_fx.shape[-1], scores.shape[1], topk_indices.shape[1], _stream(resid_fx)))


def exl3_mgemm_deferred(A, B, suh, indices, K: int, mcg: int, mul1: int, n: int, force_split: int = 0):
    """Indexed gate|up launch with a deferred epilogue (generation-4 GEMV): A [1 | bszm, m, k] fp16 raw x, B / suh int64 pointer tables, indices int64
    [bszm]; returns (slab address, S) of the slabs [bszm][n/128][S][m][128] fp32 for exl3_mgemm_act_fx."""
    _dev(A)
    _req(A.dtype == torch.half and A.dim() == 3 and A.is_contiguous(), "exl3_mgemm_deferred: A must be contiguous float16 [1 | bszm, m, k]")
    _req(B.dtype == torch.long and suh.dtype == torch.long and indices.dtype == torch.long and indices.is_contiguous(), "exl3_mgemm_deferred: int64 tables")
    bszm = indices.numel()
    _req(A.shape[0] in (1, bszm), "exl3_mgemm_deferred: A has 1 or bszm slots")
    slab = (_vp * 1)()
    S = ctypes.c_int(0)
    _check(_lib.lib().exl3_mgemm_indexed_deferred(_p(A), A.shape[0], _p(B), _p(suh), _p(indices), bszm, A.shape[1], A.shape[2], int(n), int(K),
                                                  _cb(bool(mcg), bool(mul1)), int(force_split), slab, ctypes.byref(S), _stream(A)))
    return int(slab[0]), S.value


def exl3_mgemm_act_fx(gu_slab: int, gu_S: int, gu_svh, up_off: int, B, suh, svh, indices, weights, R, K: int, mcg: int, mul1: int, k: int, m: int = 1,
                      num_tokens: int = 1, force_split: int = 0):
    """Indexed, weighted down launch of a MoE block in the fx pipeline: input silu(g) * u from exl3_mgemm_deferred's slabs, output ADDED into the int64
    fixed-point residual accumulator R [num_tokens * m, n] (see exl3_mgemm_indexed_act_fx)."""
    _dev(R)
    _req(R.dtype == torch.long and R.is_contiguous(), "exl3_mgemm_act_fx: R must be a contiguous int64 accumulator")
    _req(all(t.dtype == torch.long for t in (gu_svh, B, suh, svh, indices)) and indices.is_contiguous(), "exl3_mgemm_act_fx: int64 tables")
    _req(weights.dtype == torch.half and weights.is_contiguous() and weights.numel() == indices.numel(), "exl3_mgemm_act_fx: one fp16 weight per slot")
    _check(_lib.lib().exl3_mgemm_indexed_act_fx(gu_slab, int(gu_S), _p(gu_svh), int(up_off), _p(B), _p(suh), _p(svh), _p(indices), _p(weights), indices.numel(),
                                                _p(R), int(m), int(k), R.shape[-1], int(K), _cb(bool(mcg), bool(mul1)), int(num_tokens), int(force_split), _stream(R)))


def exl3_gemv_ex_act(gu_slabs, gu_S: int, svh_g, svh_u, B, C, suh, svh, m: int, mcg: bool, mul1: bool, flags: int = 0, force_split: int = 0,
                     c_fp32: bool = False):
    """down_proj fed by the gate / up launch's deferred slabs (gu_slabs = [gate ptr, up ptr] as returned by exl3_gemv_ex*): silu(g) * u and the
    input Hadamard happen inside this GEMV (m <= 4).  Returns (slab pointer list, S) like exl3_gemv_ex."""
    _dev(B)
    k, K = _kK(B)
    slab = (_vp * 1)()
    S = ctypes.c_int(0)
    _check(_lib.lib().exl3_gemv_ex_act(gu_slabs[0], gu_slabs[1], gu_S, _p(svh_g), _p(svh_u), _p(B), _p(C), _p(suh), _p(svh), None, m, k,
                                       B.shape[1] * 16, K, _cb(mcg, mul1), int(c_fp32), flags, force_split, slab, ctypes.byref(S), _stream(B)))
    return [int(slab[0]) if slab[0] else 0], S.value


def fx_init(x: torch.Tensor, R: torch.Tensor, ss: torch.Tensor, m: int):
    """R (int64 [m][hidden], value * 2^32) = x (fp16); ss [m][hidden/128] = block sums of squares of x: the start of the fx decode pipeline."""
    _dev(x)
    _req(x.dtype == torch.half and R.dtype == torch.int64 and ss.dtype == torch.float and x.is_contiguous() and R.is_contiguous(), "fx_init: dtypes")
    _check(_lib.lib().exl3_fx_init(_p(x), _p(R), _p(ss), m, x.shape[-1], _stream(x)))


def fx_init_prep(x: torch.Tensor, R: torch.Tensor, ss: torch.Tensor, m: int, inv_freq, positions, head_dim: int, block_table, page_size: int,
                 sin_out, cos_out, slots, attn_factor: float = 1.0):
    """fx_init and qkv_prep (the two independent set-up launches of a decode step) as ONE launch; same outputs."""
    _dev(x)
    _req(x.dtype == torch.half and R.dtype == torch.int64 and ss.dtype == torch.float and x.is_contiguous() and R.is_contiguous(), "fx_init_prep: dtypes")
    _req(sin_out.dtype == torch.float and cos_out.dtype == torch.float and sin_out.shape == cos_out.shape and sin_out.shape[-1] == 64, "fx_init_prep: sin / cos must be (m, 64) float32")
    _req(slots is None or (slots.dtype == torch.long and block_table is not None and block_table.dtype == torch.int32), "fx_init_prep: slots must be int64 and need an int32 block table")
    _req(positions.shape[0] == m, "fx_init_prep: one position per row")
    _check(_lib.lib().exl3_fx_init_prep(_p(x), _p(R), _p(ss), m, x.shape[-1], _p(inv_freq), _p(positions), float(attn_factor), int(head_dim), _p(block_table),
                                        block_table.shape[1] if block_table is not None else 0, int(page_size), _p(sin_out), _p(cos_out), _p(slots), _stream(x)))


def fx_finish_rotate(R: torch.Tensor, x, ss, norm_w, eps: float, suh, xh, xsum, m: int):
    """fx_finish + glue_rotate for one consumer (the lm_head) as ONE launch: x (optional) = fp16(R / 2^32), ss (optional) its block sums of squares,
    xh = had128(rms_norm(x) * suh) / sqrt(128) [m, hidden] fp16, xsum (optional) = block sums of xh."""
    _dev(R)
    _req(R.dtype == torch.int64 and R.is_contiguous() and xh.dtype == torch.half and xh.is_contiguous() and norm_w.dtype == torch.half and suh.dtype == torch.half,
         "fx_finish_rotate: R int64, xh / norm_w / suh float16")
    _check(_lib.lib().exl3_fx_finish_rotate(_p(R), _p(x), _p(ss), _p(norm_w), float(eps), _p(suh), _p(xh), _p(xsum), m, R.shape[-1], _stream(R)))


def fx_finish(R: torch.Tensor, x: torch.Tensor | None, ss: torch.Tensor | None, m: int):
    """x = fp16(R / 2^32), ss = its block sums of squares (either may be None)."""
    _dev(R)
    _req(R.dtype == torch.int64 and R.is_contiguous(), "fx_finish: R must be a contiguous int64 tensor")
    _check(_lib.lib().exl3_fx_finish(_p(R), _p(x), _p(ss), m, R.shape[-1], _stream(R)))


def exl3_gemv_ex_fx(R, norm_w, ss_prev, ss_out, eps: float, Bs, suhs, m: int, mcg: bool, mul1: bool, force_split: int = 0):
    """exl3_gemv_ex_norm whose residual input is the fixed-point accumulator R (int64, value * 2^32; GEMV_OUT_ATOMIC launches add into it).  The row
    scale is the PREVIOUS residual's (ss_prev); ss_out (another buffer) receives this residual's block sums of squares; finish the returned slabs
    with glue_qkv_rs / glue_act_rs / exl3_gemv_ex_act_rs (ss_prev, ss_out).  Returns (slabs, S)."""
    _dev(R)
    _req(R.dtype == torch.int64, "exl3_gemv_ex_fx: R must be int64")
    cnt = len(Bs)
    k, K = _kK(Bs[0])
    ns = (ctypes.c_int * cnt)(*[B.shape[1] * 16 for B in Bs])
    slabs = (_vp * cnt)()
    S = ctypes.c_int(0)
    _check(_lib.lib().exl3_gemv_ex_fx(_p(R), _p(norm_w), _p(ss_prev), _p(ss_out), float(eps), _parr(Bs), _parr(suhs), ns, cnt, m, k, K,
                                      _cb(mcg, mul1), force_split, slabs, ctypes.byref(S), _stream(R)))
    return [int(s) if s else 0 for s in slabs], S.value


def fx_add(R: torch.Tensor, y: torch.Tensor | None = None, slab: int = 0, S: int = 0, svh: torch.Tensor | None = None):
    """R (int64 fixed-point residual [m, hidden]) += y (fp32 [m, hidden]) or += the finished deferred slabs (slab address, S, svh) of a linear."""
    _dev(R)
    _req(R.dtype == torch.long and R.dim() == 2 and R.is_contiguous(), "fx_add: R must be a contiguous int64 [m, hidden] accumulator")
    _req((y is not None) != bool(slab), "fx_add: give y or slabs")
    _req(y is None or (y.dtype == torch.float and y.is_contiguous() and y.numel() == R.numel()), "fx_add: y must be contiguous float32 of R's shape")
    _check(_lib.lib().exl3_fx_add(_p(R), _p(y), ctypes.c_void_p(slab) if slab else None, int(S), _p(svh), R.shape[0], R.shape[1], _stream(R)))


def fx_zero_next(buf: torch.Tensor | None):
    """The NEXT generation-4 GEMV launch clears `buf` as a side job (the gate / up accumulators of the fx pipeline); None cancels."""
    if buf is None:
        _check(_lib.lib().exl3_fx_zero_next(None, 0)); return
    _dev(buf)
    _req(buf.is_contiguous() and (buf.numel() * buf.element_size()) % 16 == 0, "fx_zero_next: contiguous buffer, a multiple of 16 bytes")
    _check(_lib.lib().exl3_fx_zero_next(_p(buf), buf.numel() * buf.element_size()))


def exl3_gemv_ex_fx_atomic(R, norm_w, ss_prev, ss_out, eps: float, Bs, accs, suhs, svhs, m: int, mcg: bool, mul1: bool, force_split: int = 0) -> int:
    """exl3_gemv_ex_fx whose output rows are ADDED into the int64 fixed-point accumulators accs[i] ([m][n_i], zero on entry).  Returns S."""
    _dev(R)
    _req(R.dtype == torch.int64 and all(a.dtype == torch.int64 and a.is_contiguous() for a in accs), "exl3_gemv_ex_fx_atomic: int64 R / accumulators")
    cnt = len(Bs)
    k, K = _kK(Bs[0])
    ns = (ctypes.c_int * cnt)(*[B.shape[1] * 16 for B in Bs])
    S = ctypes.c_int(0)
    _check(_lib.lib().exl3_gemv_ex_fx_atomic(_p(R), _p(norm_w), _p(ss_prev), _p(ss_out), float(eps), _parr(Bs), _parr(accs), _parr(suhs), _parr(svhs),
                                             ns, cnt, m, k, K, _cb(mcg, mul1), force_split, ctypes.byref(S), _stream(R)))
    return S.value


def exl3_gemv_ex_actfx(g_acc, u_acc, ss_prev, ss_new, hidden: int, eps: float, B, C, suh, svh, m: int, mcg: bool, mul1: bool, flags: int = 0,
                       force_split: int = 0):
    """down_proj whose input silu(g) * u is formed from the int64 fixed-point gate / up accumulators (row-scale correction from ss_prev / ss_new);
    flags: GEMV_OUT_ATOMIC (C = the residual accumulator) or GEMV_OUT_DEFERRED.  Returns (slab pointer list, S)."""
    _dev(B)
    _req(g_acc.dtype == torch.int64 and u_acc.dtype == torch.int64, "exl3_gemv_ex_actfx: int64 accumulators")
    k, K = _kK(B)
    slab = (_vp * 1)()
    S = ctypes.c_int(0)
    _check(_lib.lib().exl3_gemv_ex_actfx(_p(g_acc), _p(u_acc), _p(ss_prev), _p(ss_new), int(hidden), float(eps), _p(B), _p(C), _p(suh), _p(svh), None,
                                         m, k, B.shape[1] * 16, K, _cb(mcg, mul1), flags, force_split, slab, ctypes.byref(S), _stream(B)))
    return [int(slab[0]) if slab[0] else 0], S.value


def exl3_gemv_ex_wpc(A, Bs, suhs, m: int, mcg: bool, mul1: bool, cpw: int, force_split: int = 0):
    """exl3_gemv_ex(raw A, deferred) in the wave-per-column-block layout (cpw column blocks of one matrix per workgroup).  Returns (slabs, S)."""
    _dev(A)
    cnt = len(Bs)
    k, K = _kK(Bs[0])
    ns = (ctypes.c_int * cnt)(*[B.shape[1] * 16 for B in Bs])
    slabs = (_vp * cnt)()
    S = ctypes.c_int(0)
    _check(_lib.lib().exl3_gemv_ex_wpc(_p(A), _parr(Bs), _parr(suhs), ns, cnt, m, k, K, _cb(mcg, mul1), int(cpw), force_split, slabs, ctypes.byref(S),
                                       _stream(A)))
    return [int(s) if s else 0 for s in slabs], S.value


def exl3_gemv_ex_resid(resid_in, norm_w, ss_prev, eps: float, prod_slab: int, prod_S: int, prod_svh, resid_out, ss_out, Bs, suhs, m: int,
                       mcg: bool, mul1: bool, force_split: int = 0, cpw: int = 4):
    """glue_resid + exl3_gemv_ex_norm in one launch (m <= 4): the residual add of the producer linear (its deferred slabs prod_slab / prod_S +
    prod_svh) is finished inside this GEMV; column-block-0 workgroups write resid_out (a different buffer than resid_in) and ss_out.  The RMSNorm
    scale applied here is resid_in's (ss_prev); finish this launch's slabs with glue_qkv_rs / exl3_gemv_ex_act_rs.  Returns (slabs, S)."""
    _dev(resid_in)
    cnt = len(Bs)
    k, K = _kK(Bs[0])
    ns = (ctypes.c_int * cnt)(*[B.shape[1] * 16 for B in Bs])
    slabs = (_vp * cnt)()
    S = ctypes.c_int(0)
    _check(_lib.lib().exl3_gemv_ex_resid(_p(resid_in), _p(norm_w), _p(ss_prev), float(eps), prod_slab, prod_S, _p(prod_svh), _p(resid_out), _p(ss_out),
                                         _parr(Bs), _parr(suhs), ns, cnt, m, k, K, _cb(mcg, mul1), int(cpw), force_split, slabs, ctypes.byref(S),
                                         _stream(resid_in)))
    return [int(s) if s else 0 for s in slabs], S.value


def exl3_gemv_ex_act_rs(gu_slabs, gu_S: int, svh_g, svh_u, ss_prev, ss_new, hidden: int, eps: float, B, C, suh, svh, m: int, mcg: bool, mul1: bool,
                        flags: int = 0, force_split: int = 0, c_fp32: bool = False, cpw: int = 0):
    """exl3_gemv_ex_act for gate / up slabs written by exl3_gemv_ex_resid (row scale correction from ss_prev / ss_new)."""
    _dev(B)
    k, K = _kK(B)
    slab = (_vp * 1)()
    S = ctypes.c_int(0)
    _check(_lib.lib().exl3_gemv_ex_act_rs(gu_slabs[0], gu_slabs[1], gu_S, _p(svh_g), _p(svh_u), _p(ss_prev), _p(ss_new), int(hidden), float(eps),
                                          _p(B), _p(C), _p(suh), _p(svh), None, m, k, B.shape[1] * 16, K, _cb(mcg, mul1), int(c_fp32), flags,
                                          int(cpw), force_split, slab, ctypes.byref(S), _stream(B)))
    return [int(slab[0]) if slab[0] else 0], S.value


def attn_prefill_paged(q, out, k_pages, v_pages, block_table, cache_seqlens, scale: float | None = None):
    """Causal attention of a q_len-token chunk over paged fp16 K/V (the pages dequant_cache_paged expands, or an fp16 cache): q / out
    (bsz, q_len, heads_q, head_dim) fp16, k_pages / v_pages (pages, page, heads_kv, head_dim) fp16, cache_seqlens int32 (bsz) INCLUDING the chunk
    (append first, as the reference does).  The reference's flash_attn_with_kvcache(..., causal=True) step of the prefill path."""
    _dev(q)
    _req(q.dtype == torch.half and out.dtype == torch.half and q.shape == out.shape and q.dim() == 4 and out.is_contiguous(),
         "attn_prefill: q / out must be float16 (bsz, q_len, heads, head_dim), out contiguous")
    _req(q.stride(3) == 1 and q.stride(2) == q.shape[3] and q.stride(0) == q.shape[1] * q.stride(1) and q.stride(1) % 8 == 0,
         "attn_prefill: q may only be strided between tokens (a column range of a wider matrix)")
    _req(k_pages.dtype == torch.half and v_pages.dtype == torch.half and k_pages.shape == v_pages.shape and k_pages.dim() == 4
         and k_pages.is_contiguous() and v_pages.is_contiguous(), "attn_prefill: k / v pages must be contiguous float16 (pages, page, heads_kv, head_dim)")
    _req(block_table.dtype == torch.int32 and cache_seqlens.dtype == torch.int32 and block_table.is_contiguous(), "attn_prefill: block_table / cache_seqlens must be int32")
    bsz, q_len, hq, hd = q.shape
    _req(k_pages.shape[3] == hd and block_table.shape[0] == bsz, "attn_prefill: shape mismatch")
    _check(_lib.lib().exl3_attn_prefill_paged_strided(_p(q), q.stride(1), _p(out), _p(k_pages), _p(v_pages), _p(block_table), _p(cache_seqlens), bsz, q_len, hq,
                                              k_pages.shape[2], hd, block_table.shape[1], k_pages.shape[1],
                                              float(scale if scale is not None else hd ** -0.5), _stream(q)))


def attn_decode_qcache_split(q, k_cache, k_scales, v_cache, v_scales, block_table, cache_seqlens, max_len: int, workspace: torch.Tensor,
                             scale: float | None = None) -> int:
    """The context-split half of attn_decode_qcache (head_dim 128): partial records stay in `workspace`; returns the split count for
    exl3_gemv_ex_attm, which merges them inside o_proj's launch."""
    _dev(q)
    _req(q.dtype == torch.half and q.dim() == 3 and q.shape[-1] in (64, 128) and q.is_contiguous(), "attn_decode_split: q must be contiguous (bsz, heads, 64 | 128) float16")
    _req(block_table.dtype == torch.int32 and cache_seqlens.dtype == torch.int32, "attn_decode: block_table / cache_seqlens must be int32")
    _req(workspace is not None and workspace.dtype == torch.float and workspace.is_contiguous(), "attn_decode_split: float32 workspace required")
    bsz, hq, hd = q.shape
    hkv = k_scales.shape[-1] * 32 // hd
    kb, vb = _kv_bits(k_cache, k_scales), _kv_bits(v_cache, v_scales)
    ns = ctypes.c_int(0)
    _check(_lib.lib().exl3_attn_decode_qcache_split(_p(q), _p(k_cache), _p(k_scales), _p(v_cache), _p(v_scales), _p(block_table), _p(cache_seqlens),
                                                    bsz, block_table.shape[1], k_cache.shape[1], kb, vb, hq, hkv, hd, int(max_len),
                                                    float(scale if scale is not None else hd ** -0.5), _p(workspace), workspace.numel(), ctypes.byref(ns), _stream(q)))
    return ns.value


def attn_decode_qcache_split_qkv(slabs, S: int, svh_q, svh_k, svh_v, q_out, inv_freq, positions, k_cache, k_scales, v_cache, v_scales, block_table,
                                 cache_seqlens, max_len: int, workspace: torch.Tensor, tab, ss_prev, ss_new, hidden: int, eps: float,
                                 rope_mode: int = 2, attn_factor: float = 1.0, scale: float | None = None):
    """glue_qkv_rs(tab=...) + attn_decode_qcache_split as one launch (libtorch/attention.cpp:386-440 as one node): the context-split kernel finishes
    its own query heads from the q|k|v launch's slabs and the split that holds the new token appends its K / V.  cache_seqlens include the new token.
    Returns (splits, fused): `fused` False = the two-launch form ran (shapes the matrix-pipe split kernel does not take); same results either way."""
    _dev(q_out)
    _req(q_out.dtype == torch.half and q_out.dim() == 3 and q_out.shape[-1] in (64, 128) and q_out.is_contiguous(), "attn_decode_split_qkv: q_out must be contiguous (bsz, heads, 128 or 64) float16")
    _req(block_table.dtype == torch.int32 and cache_seqlens.dtype == torch.int32, "attn_decode: block_table / cache_seqlens must be int32")
    _req(workspace is not None and workspace.dtype == torch.float and workspace.is_contiguous(), "attn_decode_split_qkv: float32 workspace required")
    _req(tab is not None and len(tab) == 3, "attn_decode_split_qkv: needs the (sin, cos, slots) tables of qkv_prep")
    bsz, hq, hd = q_out.shape
    hkv = k_scales.shape[-1] * 32 // hd
    kb, vb = _kv_bits(k_cache, k_scales), _kv_bits(v_cache, v_scales)
    ns, fused = ctypes.c_int(0), ctypes.c_int(0)
    _check(_lib.lib().exl3_attn_decode_qcache_split_qkv(slabs[0], slabs[1], slabs[2], S, _p(svh_q), _p(svh_k), _p(svh_v), _p(q_out), _p(inv_freq), _p(positions),
                                                        float(attn_factor), int(rope_mode), _p(ss_prev), _p(ss_new), int(hidden), float(eps), _p(tab[0]), _p(tab[1]),
                                                        _p(tab[2]), _p(k_cache), _p(k_scales), _p(v_cache), _p(v_scales), _p(block_table), _p(cache_seqlens), bsz,
                                                        block_table.shape[1], k_cache.shape[1], kb, vb, hq, hkv, hd, int(max_len),
                                                        float(scale if scale is not None else hd ** -0.5), _p(workspace), workspace.numel(), ctypes.byref(ns),
                                                        ctypes.byref(fused), _stream(q_out)))
    return ns.value, bool(fused.value)


def exl3_gemv_ex_attm(part: torch.Tensor, nsplit: int, heads_q: int, heads_kv: int, B, C, suh, svh, m: int, mcg: bool, mul1: bool, flags: int = 0,
                      force_split: int = 0, c_fp32: bool = False, head_dim: int = 128):
    """o_proj whose input is the decode attention's output, merged from the context-split partial records inside the launch (ext.attn_decode_qcache_split).
    head_dim 128, or 64 (two query heads per Hadamard block of o_proj's input, at most 16 splits).  Returns ([slab], S) like exl3_gemv_ex."""
    _dev(B)
    k, K = _kK(B)
    _req(head_dim in (64, 128) and k == heads_q * head_dim and heads_q % heads_kv == 0 and (heads_kv * head_dim) % 128 == 0, "exl3_gemv_ex_attm: k must be heads_q x head_dim")
    slab = (_vp * 1)()
    S = ctypes.c_int(0)
    _check(_lib.lib().exl3_gemv_ex_attm(_p(part), int(nsplit), heads_q // heads_kv, heads_kv * head_dim // 128, int(head_dim), _p(B), _p(C), _p(suh), _p(svh), None, m, k, B.shape[1] * 16, K,
                                        _cb(mcg, mul1), int(c_fp32), flags, force_split, slab, ctypes.byref(S), _stream(B)))
    return [int(slab[0]) if slab[0] else 0], S.value


def exl3_gemv_ex_qkvm(slabs, S_qkv: int, svh_q, svh_k, svh_v, tab, ss_prev, ss_new, hidden: int, eps: float, head_dim: int, heads_kv: int,
                      k_cache, k_scales, v_cache, v_scales, B, C, suh, svh, m: int, mcg: bool, mul1: bool, flags: int = 0, force_split: int = 0,
                      rope_mode: int = 2, q_out=None, c_fp32: bool = False):
    """o_proj fed straight by the q|k|v launch's deferred slabs (the decode step without the attention core): glue_qkv_rs(tab=...)'s work -- reduce,
    output Hadamard, row-scale correction, svh, RoPE, 4-bit K / V append -- runs inside o_proj's launch (same bits, one launch less per layer).
    tab = (sin, cos, slots) of qkv_prep.  Returns ([slab], S) like exl3_gemv_ex."""
    _dev(B)
    k, K = _kK(B)
    _req(tab is not None and len(tab) == 3, "exl3_gemv_ex_qkvm: needs the (sin, cos, slots) tables of qkv_prep")
    _req(_kv_bits(k_cache, k_scales) == 4 and _kv_bits(v_cache, v_scales) == 4, "exl3_gemv_ex_qkvm: 4-bit K and V cache")
    _req(head_dim in (64, 128) and k % 128 == 0, "exl3_gemv_ex_qkvm: head_dim 64 | 128")
    slab = (_vp * 1)()
    S = ctypes.c_int(0)
    _check(_lib.lib().exl3_gemv_ex_qkvm(slabs[0], slabs[1], slabs[2], int(S_qkv), _p(svh_q), _p(svh_k), _p(svh_v), _p(tab[0]), _p(tab[1]), _p(tab[2]),
                                        _p(ss_prev), _p(ss_new), int(hidden), float(eps), int(rope_mode), int(head_dim), int(heads_kv), _p(q_out),
                                        _p(k_cache), _p(k_scales), _p(v_cache), _p(v_scales), _p(B), _p(C), _p(suh), _p(svh), None, m, k, B.shape[1] * 16, K,
                                        _cb(mcg, mul1), int(c_fp32), flags, force_split, slab, ctypes.byref(S), _stream(B)))
    return [int(slab[0]) if slab[0] else 0], S.value


def attn_decode_qcache(q, out, k_cache, k_scales, v_cache, v_scales, block_table, cache_seqlens, max_len: int, scale: float | None = None,
                       workspace: torch.Tensor | None = None, sinks: torch.Tensor | None = None):
    """Decode attention straight from the quantized paged cache.  q / out: (bsz, heads_q, 128) fp16; caches (pages, page, G * bits) int32 +
    scales (pages, page, G) fp16 as written by quant_cache_paged / glue_qkv; cache_seqlens int32 (bsz) INCLUDING the new token."""
    _dev(q)
    _req(q.dtype == torch.half and out.dtype == torch.half and q.shape == out.shape and q.dim() == 3 and q.shape[-1] in (64, 128),
         "attn_decode: q/out must be (bsz, heads, 64 | 128) float16")
    _req(q.is_contiguous() and out.is_contiguous(), "attn_decode: q/out must be contiguous")
    _req(block_table.dtype == torch.int32 and cache_seqlens.dtype == torch.int32, "attn_decode: block_table / cache_seqlens must be int32")
    bsz, hq, hd = q.shape
    G = k_scales.shape[-1]
    hkv = G * 32 // hd
    kb, vb = _kv_bits(k_cache, k_scales), _kv_bits(v_cache, v_scales)
    nsplit_max = (max_len + 31) // 32
    need = bsz * (hq * hd // 128) * nsplit_max * 132              # one record per (sequence, 128-value block, query index) and split
    if workspace is None and (nsplit_max > 1 or sinks is not None):
        workspace = torch.empty((need,), dtype=torch.float, device=q.device)
    if sinks is not None:
        # learned per-head sink logits (float32 [heads_q]): in the softmax denominator only (triton_paged.py:1030-1050)
        _req(sinks.dtype == torch.float and sinks.numel() == hq and sinks.is_contiguous(), "attn_decode: sinks must be float32 [heads_q]")
        _check(_lib.lib().exl3_attn_decode_qcache_sinks(_p(q), _p(out), _p(k_cache), _p(k_scales), _p(v_cache), _p(v_scales), _p(block_table), _p(cache_seqlens),
                                                        bsz, block_table.shape[1], k_cache.shape[1], kb, vb, hq, hkv, hd, int(max_len),
                                                        float(scale if scale is not None else hd ** -0.5), _p(workspace), workspace.numel(), _p(sinks), _stream(q)))
        return
    _check(_lib.lib().exl3_attn_decode_qcache(_p(q), _p(out), _p(k_cache), _p(k_scales), _p(v_cache), _p(v_scales), _p(block_table), _p(cache_seqlens),
                                              bsz, block_table.shape[1], k_cache.shape[1], kb, vb, hq, hkv, hd, int(max_len),
                                              float(scale if scale is not None else hd ** -0.5), _p(workspace),
                                              workspace.numel() if workspace is not None else 0, _stream(q)))


# ---- persistent decode step (generation 5, exl3_pstep.hip) ------------------------------------------------------------------------------
class PersistentStep:
    """One launch for every quantized linear of a batch-1 decode step (all layers + lm_head) and the glue between them.  Replaces the per-layer graphs of
    exllamav3_ext/libtorch/attention.cpp:246-330 (attention core excluded) + libtorch/mlp.cpp:14-91; the plan (rectangles per CU, slab / row buffers)
    is built once from the layer tensors.  `layers`: dicts with LinearEXL3-like entries q, k, v, o, gate, up, down (attributes trellis, suh, svh, K) and
    tensors norm1, norm2, kcache = (words, scales), vcache = (words, scales).  4-bit cache, hidden <= 4096; one codebook for the model (mul1 | 3INST | mcg), the tensors
    of one fused linear (q / k / v; gate / up) share their bits per weight, the layers' linears have one width or two adjacent ones (a fractional-bpw checkpoint), the
    lm_head the layers' or 6 bits -- exl3_pstep_create checks and names the kernel instantiations (exl3_pstep.kspec.hip)."""

    def __init__(self, layers, head, final_norm, hidden: int, heads_q: int, heads_kv: int, head_dim: int, eps: float, rope_mode: int = 2, stamps: bool = False,
                 attention: bool = False, repack: bool | None = None, tp: tuple | None = None):
        """attention: the decode attention over the 4-bit paged cache runs INSIDE the step (o_proj's preparation: one (kv head, context split) item per CU, the partial
        records merged by the consumers; libtorch/attention.cpp:246-504 at q_len 1) -- run() then needs block_table / cache_seqlens.  head_dim 128.
        repack: None (default) / True = the plan copies every op's packed words once into the order its streaming waves read them (a second copy of the weights owned by
        the plan: one contiguous run per wave; SURVEY 8(f)4's legal load-time transform); False = stream the caller's checkpoint-layout tensors as they are.
        tp = (ranks, this rank): the layers are this rank's tensor-parallel shards (q / k / v / gate / up column shards, o / down row shards, an lm_head column shard;
        heads_q / heads_kv the rank's, hidden the model's) and the all-reduce behind o_proj / down_proj (model/model_tp_backend.py:119-126) happens INSIDE the step: every
        rank pushes its partial rows into every rank's exchange buffer -- exchange tp_handle() over the process group, tp_open_peer() each, tp_commit(), barrier."""
        def lin(l):
            t = l.trellis
            _req(t.dim() == 3 and l.suh is not None and l.svh is not None, "PersistentStep: EXL3 linears with suh / svh")
            return _lib.PstepLinear(_p(t), _p(l.suh), _p(l.svh), t.shape[0] * 16, t.shape[1] * 16, t.shape[-1] // 16, 1 if l.mcg else (2 if l.mul1 else 0))
        K = min(L[name].trellis.shape[-1] // 16 for L in layers for name in ("q", "k", "v", "o", "gate", "up", "down"))
        cb = 1 if layers[0]["q"].mcg else (2 if layers[0]["q"].mul1 else 0)
        arr = (_lib.PstepLayer * len(layers))()
        for i, L in enumerate(layers):
            for name in ("q", "k", "v", "o", "gate", "up", "down"):
                setattr(arr[i], name, lin(L[name]))
            arr[i].norm1, arr[i].norm2 = _p(L["norm1"]), _p(L["norm2"])
            (kw, ks), (vw, vs) = L["kcache"], L["vcache"]
            arr[i].k_cache, arr[i].k_scales, arr[i].v_cache, arr[i].v_scales = _p(kw), _p(ks), _p(vw), _p(vs)
        KH = head.trellis.shape[-1] // 16
        hl = lin(head)
        _dev(final_norm)
        self._h = ctypes.c_void_p(None)
        self._keep = (layers, head, final_norm)          # the plan holds raw pointers
        _check(_lib.lib().exl3_pstep_create(ctypes.byref(self._h), arr, len(layers), ctypes.byref(hl), _p(final_norm), int(hidden), int(heads_q), int(heads_kv),
                                            int(head_dim), int(K), int(cb), float(eps), int(rope_mode),
                                            (1 if stamps else 0) | (4 if attention else 0) | ((KH << 8) if KH != K else 0) | (16 if repack is False else 0)
                                            | (((int(tp[0]) << 12) | (int(tp[1]) << 16)) if tp is not None and int(tp[0]) > 1 else 0)))
        self.tp = tuple(int(v) for v in tp) if tp is not None and int(tp[0]) > 1 else None
        self.n_layers = len(layers)
        self.attention = bool(attention)
        self.head_dim = int(head_dim)

    def run(self, R: torch.Tensor, logits: torch.Tensor, q_out: torch.Tensor | None, rope_sin: torch.Tensor, rope_cos: torch.Tensor, slots: torch.Tensor,
            block_table: torch.Tensor | None = None, cache_seqlens: torch.Tensor | None = None, page_size: int = 0, scale: float | None = None):
        """R: int64 fixed-point residual of the embedded token (fx_init / fx_init_prep, which also fill rope_sin / rope_cos / slots); graph-capturable.
        With the attention inside: block_table int32 (1, pages) and cache_seqlens int32 (1,) = the length INCLUDING the new token (read on the device)."""
        _dev(R)
        _req(R.dtype == torch.int64 and logits.dtype == torch.half and slots.dtype == torch.int64, "PersistentStep.run: dtypes")
        if self.attention:
            _req(block_table is not None and cache_seqlens is not None and block_table.dtype == torch.int32 and cache_seqlens.dtype == torch.int32 and page_size > 0,
                 "PersistentStep.run: the plan has the attention inside: block_table / cache_seqlens (int32) and the page size")
            _check(_lib.lib().exl3_pstep_run_attn(self._h, _p(R), _p(logits), _p(q_out), _p(rope_sin), _p(rope_cos), _p(slots), _p(block_table), _p(cache_seqlens),
                                                  int(block_table.shape[-1]), int(page_size), float(scale if scale is not None else self.head_dim ** -0.5), _stream(R)))
            return
        _check(_lib.lib().exl3_pstep_run(self._h, _p(R), _p(logits), _p(q_out), _p(rope_sin), _p(rope_cos), _p(slots), _stream(R)))

    def error(self) -> bool:
        """Synchronises; True if an edge / tagged line ever timed out since the last query (results invalid)."""
        return bool(_check(_lib.lib().exl3_pstep_error(self._h, torch.cuda.current_stream().cuda_stream)))

    def error_peek(self) -> bool:
        """No synchronisation: True once the kernel has reported a timed-out wait to the pinned host word (some EARLIER step wrote NaN logits); error() confirms and clears."""
        return bool(_check(_lib.lib().exl3_pstep_error_peek(self._h)))

    def attn_geometry(self, length: int):
        """(context splits in use, tokens per split, 128-token steps per split) of the attention inside the step at a sequence length (incl. the new token)."""
        out = (ctypes.c_int * 3)()
        _check(_lib.lib().exl3_pstep_attn_geometry(self._h, int(length), out))
        return int(out[0]), int(out[1]), int(out[2])

    def unpack_op(self, op: int, mat: int, like: torch.Tensor) -> torch.Tensor:
        """The plan's repacked words of matrix `mat` of op `op` (4 * layer + {0 q|k|v, 1 o, 2 gate|up, 3 down}; last op = lm_head) copied back into a checkpoint-layout
        tensor shaped like `like` (the original trellis): equal to it bit for bit if the load-time permutation is one."""
        out = torch.zeros_like(like)
        _check(_lib.lib().exl3_pstep_unpack_op(self._h, int(op), int(mat), _p(out), _stream(like)))
        return out

    def tp_handle(self) -> bytes:
        """The 64-byte IPC handle of this rank's exchange buffer (a tensor-parallel plan)."""
        buf = ctypes.create_string_buffer(64)
        _check(_lib.lib().exl3_pstep_tp_handle(self._h, buf))
        return bytes(buf.raw)

    def tp_open_peer(self, rank: int, handle: bytes):
        _check(_lib.lib().exl3_pstep_tp_open_peer(self._h, int(rank), ctypes.create_string_buffer(handle, 64)))

    def tp_commit(self):
        _check(_lib.lib().exl3_pstep_tp_commit(self._h))

    def tp_peek(self):
        """Diagnostics: this rank's exchange buffer (the partial lines of every rank's row shards) as a numpy uint32 array; synchronises the device."""
        import numpy as np
        buf = (ctypes.c_uint32 * (1 << 23))()
        n = _lib.lib().exl3_pstep_tp_peek(self._h, buf, 4 << 23)
        if n < 0:
            raise RuntimeError(_lib.last_error())
        return np.frombuffer(buf, dtype=np.uint32, count=int(n) // 4).copy()

    def set(self, decode_ahead_units: int = -1, spin_limit: int = 0):
        _check(_lib.lib().exl3_pstep_set(self._h, int(decode_ahead_units), int(spin_limit)))

    def describe(self) -> str:
        buf = ctypes.create_string_buffer(2048)
        _check(_lib.lib().exl3_pstep_describe(self._h, buf, 2048))
        return buf.value.decode()

    def stamps(self):
        """Phase stamps of the last run as a numpy array [ops][CUs][32] (100 MHz ticks); empty unless created with stamps=True (tools/pstep_stamps.py)."""
        import numpy as np
        n = (4 * self.n_layers + 1) * 1024 * 32
        buf = (ctypes.c_uint64 * n)()
        got = _lib.lib().exl3_pstep_stamps(self._h, buf, n, torch.cuda.current_stream().cuda_stream)
        if got < 0:
            raise RuntimeError(_lib.last_error())
        return np.frombuffer(buf, dtype=np.uint64, count=int(got)).reshape(4 * self.n_layers + 1, -1, 32).copy()

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                _lib.lib().exl3_pstep_destroy(self._h)
                self._h = ctypes.c_void_p(None)
        except Exception:
            pass
