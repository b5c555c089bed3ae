"""
CPU ORACLE (test infrastructure, NOT product code) for the EXL3 quantized-linear hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (exllamav3_amd/) never imports it and fails loudly without the HIP library.

Every function is a numpy restatement of the reference algorithm, citing the reference
file:line it follows (paths relative to /root/reference/exllamav3/).  Integer / byte /
index work is bit-exact by construction; floating-point rounding points follow the reference
kernels.

Parity pinning (see oracle/README.md and tests/test_oracle_pins.py):
  * tensor_core_perm, Hadamard matrices, RMSNorm, RoPE, unpack_bf, and the constructive
    tail-biting trellis generator are pinned against fixtures produced by IMPORTING the
    reference's own Python (tests/golden/make_golden.py, run in the build container).
  * trellis bit-window extraction + tile permutation + mul1 codebook + both Hadamards + suh/svh
    are pinned against the reference's own CPU implementation (exllamav3_ext/cpu/moe_mul1.cpp,
    compiled from where it lies into oracle/_ref/ by oracle/build_ref.sh).
  * all three codebooks for all 65536 states incl. the single fp16 rounding (decode_3inst / decode_3inst_2, quant/codebook.cuh:56-123),
    the trellis window readers for K = 1..8 incl. every dq_dispatch fast path (quant/exl3_dq.cuh:15-293), and the KV-cache quantizer
    (words, scales, dequantized values for 2..8 bits, ragged groups, paged addressing: cache/q_cache_kernels.cuh:61-399) are pinned BIT
    FOR BIT against the reference's own device headers compiled for the host from where they lie (oracle/cuda_host_shim.h +
    oracle/ref_cuda_harness.cpp -> oracle/_ref/libexl3_ref_cuda.so).
  * the MoE router is pinned against the reference's torch routing (modules/block_sparse_mlp.py:95-127, fixture).
  * 3INST codebook, additionally: the reference constant codebook_scale = 1.24371088
    (modules/quant/exl3_lib/quantize.py:16) = std of the codebook over all 65536 states.
"""
from __future__ import annotations
import numpy as np

# ------------------------------------------------------------------------------------------
# Format constants
# ------------------------------------------------------------------------------------------

CB_3INST, CB_MCG, CB_MUL1 = 0, 1, 2
MCG_MULT = 0xCBAC1FED      # modules/quant/exl3_lib/quantize.py:18
MUL1_MULT = 0x83DCD12D     # modules/quant/exl3_lib/quantize.py:19
HAD_SCALE_128 = np.float32(0.088388347648)  # exllamav3_ext/quant/hadamard.cu:103


def cb_index(mcg: bool, mul1: bool) -> int:
    """exllamav3_ext/quant/reconstruct.cu:128-130: mcg wins over mul1."""
    return CB_MCG if mcg else (CB_MUL1 if mul1 else CB_3INST)


def tensor_core_perm() -> np.ndarray:
    """perm[t] = row*16 + col of stream index t in the 16x16 tile (row = k offset, col = n offset).
    modules/quant/exl3_lib/quantize.py:21-44, exllamav3_ext/cpu/moe_mul1.cpp:97-111."""
    t = np.arange(256)
    l, j = t // 8, t % 8
    row = 2 * (l % 4) + (j & 1) + 8 * ((j >> 1) & 1)
    col = l // 4 + 8 * (j >> 2)
    return (row * 16 + col).astype(np.int32)


# ------------------------------------------------------------------------------------------
# Bitstream pack / unpack     exllamav3_ext/quant/pack.cu:9-138, quant/exl3_dq.cuh:15-31
# ------------------------------------------------------------------------------------------

def _tile_words(trellis: np.ndarray) -> np.ndarray:
    """(kt, nt, 16K) int16 -> (kt, nt, 8K) uint32 little-endian words (pack.cu:111-112)."""
    t = np.ascontiguousarray(trellis).view(np.uint16)
    return t.view(np.uint32)


def unpack_trellis(trellis: np.ndarray, K: int) -> np.ndarray:
    """Packed tiles (kt, nt, 16K) int16 -> 16-bit trellis states (kt, nt, 256) uint16 in STREAM order.
    State of weight t = bits [(t+1)K-16, (t+1)K) of the circular MSB-first tile bitstream.
    exllamav3_ext/quant/exl3_dq.cuh:15-31 (dq), quant/pack.cu:97-138 (unpack_trellis_kernel),
    cpu/moe_mul1.cpp:162-172 (decode_state_scalar)."""
    assert trellis.shape[-1] == 16 * K
    w = _tile_words(trellis).astype(np.uint64)                       # (kt, nt, 8K)
    nw = 8 * K
    t = np.arange(256, dtype=np.int64)
    b0 = t * K + K - 16 + 256 * K
    b1 = b0 + 16
    i0 = (b0 // 32) % nw
    i1 = ((b1 - 1) // 32) % nw
    sh = (((b1 - 1) // 32) + 1) * 32 - b1
    merged = (w[..., i0] << np.uint64(32)) | w[..., i1]
    return ((merged >> sh.astype(np.uint64)) & np.uint64(0xFFFF)).astype(np.uint16)


def pack_trellis(states: np.ndarray, K: int) -> np.ndarray:
    """States (kt, nt, 256) uint16 (only the low K bits of each are stored) -> packed (kt, nt, 16K) int16.
    Symbols are concatenated MSB-first into 16-bit chunks; chunk c is stored at int16 index c^1 (SWAP16).
    exllamav3_ext/quant/pack.cu:9-57."""
    s = (states.astype(np.uint64) & np.uint64((1 << K) - 1))         # (kt, nt, 256)
    kt, nt, _ = s.shape
    # bit p of the tile stream (p = t*K + b, b = 0 is the symbol's MSB)
    bits = ((s[..., :, None] >> np.arange(K - 1, -1, -1, dtype=np.uint64)) & np.uint64(1)).astype(np.uint8)
    bits = bits.reshape(kt, nt, 256 * K)
    chunks = bits.reshape(kt, nt, 16 * K, 16)
    weights = (1 << np.arange(15, -1, -1)).astype(np.uint32)
    vals = (chunks.astype(np.uint32) * weights).sum(-1).astype(np.uint16)   # (kt, nt, 16K) chunk order
    out = np.empty_like(vals)
    out[..., 0::2] = vals[..., 1::2]
    out[..., 1::2] = vals[..., 0::2]
    return out.view(np.int16)


def random_tailbiting_states(rng: np.random.Generator, shape, K: int) -> np.ndarray:
    """Constructive valid tail-biting state sequence: state_t = sym_t | sym_{t-1} << K | ... (16 bits).
    Restates tests/test_quant_fn.py:100-113."""
    sym = rng.integers(0, 1 << K, size=tuple(shape) + (256,), dtype=np.uint64)
    st = np.zeros_like(sym)
    n = (16 + K - 1) // K
    for i in range(n):
        st |= np.roll(sym, i, axis=-1) << np.uint64(i * K)
    return (st & np.uint64(0xFFFF)).astype(np.uint16)


def pack_signs(signs: np.ndarray) -> np.ndarray:
    """fp16 vector (multiple of 16) -> int16 bitfield, bit b of word w = 1 iff element 16w+b is negative.
    exllamav3_ext/quant/pack.cu:177-201; inverse = modules/quant/exl3.py:142-158 (unpack_bf)."""
    neg = (np.asarray(signs).view(np.uint16) >> 15).astype(np.uint32).reshape(-1, 16)
    return (neg << np.arange(16, dtype=np.uint32)).sum(-1).astype(np.uint16).view(np.int16)


def unpack_bf(bitfield: np.ndarray) -> np.ndarray:
    """modules/quant/exl3.py:142-158."""
    b = np.asarray(bitfield).view(np.uint16).astype(np.uint32)
    ex = ((b[:, None] >> np.arange(16, dtype=np.uint32)) & 1).reshape(-1)
    return (1.0 - ex.astype(np.float16) * np.float16(2.0)).astype(np.float16)


# ------------------------------------------------------------------------------------------
# Codebooks      exllamav3_ext/quant/codebook.cuh:56-90
# ------------------------------------------------------------------------------------------

def decode(states: np.ndarray, cb: int) -> np.ndarray:
    """16-bit state -> fp16 value.  All integer arithmetic mod 2^32; exactly ONE fp16 rounding per weight
    (cb0/cb1: one RN fp16 add; cb2: one RN fp16 fma), evaluated here in float64 (exact) then rounded once."""
    s = states.astype(np.uint64)
    if cb == CB_3INST:
        x = (s * np.uint64(89226354) + np.uint64(64248484)) & np.uint64(0xFFFFFFFF)
    elif cb == CB_MCG:
        x = (s * np.uint64(MCG_MULT)) & np.uint64(0xFFFFFFFF)
    elif cb == CB_MUL1:
        x = (s * np.uint64(MUL1_MULT)) & np.uint64(0xFFFFFFFF)
        b = (x & np.uint64(0xFF)) + ((x >> np.uint64(8)) & np.uint64(0xFF)) + \
            ((x >> np.uint64(16)) & np.uint64(0xFF)) + (x >> np.uint64(24))
        h = (np.uint64(0x6400) + b).astype(np.uint16).view(np.float16).astype(np.float64)   # = 1024 + b exactly
        k_inv = np.array([0x1EEE], dtype=np.uint16).view(np.float16).astype(np.float64)[0]
        k_bias = np.array([0xC931], dtype=np.uint16).view(np.float16).astype(np.float64)[0]
        return (h * k_inv + k_bias).astype(np.float16)            # exact in f64, single rounding = hfma
    else:
        raise ValueError(cb)
    x = (x & np.uint64(0x8FFF8FFF)) ^ np.uint64(0x3B603B60)
    lo = (x & np.uint64(0xFFFF)).astype(np.uint16).view(np.float16).astype(np.float64)
    hi = (x >> np.uint64(16)).astype(np.uint16).view(np.float16).astype(np.float64)
    return (lo + hi).astype(np.float16)                            # exact in f64, single rounding = hadd


def decode_split(states: np.ndarray, cb: int):
    """cb0/cb1 only: the two fp16 halves whose (unrounded) sum is the weight."""
    s = states.astype(np.uint64)
    if cb == CB_3INST:
        x = (s * np.uint64(89226354) + np.uint64(64248484)) & np.uint64(0xFFFFFFFF)
    else:
        x = (s * np.uint64(MCG_MULT)) & np.uint64(0xFFFFFFFF)
    x = (x & np.uint64(0x8FFF8FFF)) ^ np.uint64(0x3B603B60)
    lo = (x & np.uint64(0xFFFF)).astype(np.uint16).view(np.float16)
    hi = (x >> np.uint64(16)).astype(np.uint16).view(np.float16)
    return lo, hi


def reconstruct(trellis: np.ndarray, K: int, cb: int, n_offset: int = 0, n_size: int | None = None) -> np.ndarray:
    """Packed (k/16, n/16, 16K) -> W_hat (k, n) fp16, rotated basis.  exllamav3_ext/quant/reconstruct.cu:13-84,98-144."""
    if n_size is None:
        n_size = trellis.shape[1] * 16 - n_offset
    assert n_offset % 128 == 0 and n_size % 128 == 0
    tr = trellis[:, n_offset // 16:(n_offset + n_size) // 16]
    kt, nt, _ = tr.shape
    vals = decode(unpack_trellis(tr, K), cb)                       # (kt, nt, 256) stream order
    tile = np.empty_like(vals)
    tile[..., tensor_core_perm()] = vals                          # tile[perm[t]] = v[t]  (moe_mul1.cpp:1103)
    return tile.reshape(kt, nt, 16, 16).transpose(0, 2, 1, 3).reshape(kt * 16, nt * 16)


# ------------------------------------------------------------------------------------------
# 128-point Hadamard       exllamav3_ext/quant/hadamard_inner.cuh:17-279, quant/hadamard.cu:88-173
# ------------------------------------------------------------------------------------------

def _fwht128_f32(v: np.ndarray) -> np.ndarray:
    """In-place-style fp32 FWHT over the last axis (length 128), Sylvester/natural order, stage order
    bit0, bit1 (in-lane H4, hadamard_inner.cuh:117-129) then bits 2..6 (shuffle_had_f4x32 :17-45)."""
    v = v.astype(np.float32).copy()
    shp = v.shape
    v = v.reshape(-1, 128)
    idx = np.arange(128)
    for bit in (1, 2, 4, 8, 16, 32, 64):
        p = v[:, idx ^ bit]
        sgn = np.where(idx & bit, np.float32(-1), np.float32(1)).astype(np.float32)
        v = (v * sgn + p).astype(np.float32)
    return v.reshape(shp)


def had_r_128(x: np.ndarray, pre_scale=None, post_scale=None, scale: float = 1.0, out_dtype=None) -> np.ndarray:
    """y = (x.view(-1,128) @ H128) * scale/sqrt(128), optional fp16 pre/post scale vectors.
    fp16 input: pre-scale multiply in fp16, butterflies fp32, round to fp16, post-scale multiply in fp16
    (hadamard_inner.cuh:93-147).  fp32 input: everything fp32 (:151-208); fp32->fp16 variant rounds before
    the fp16 post-scale (:212-279)."""
    rows, cols = x.shape
    assert cols % 128 == 0
    r_scale = np.float32(np.float32(scale) * HAD_SCALE_128)
    if x.dtype == np.float16:
        v = x
        if pre_scale is not None:
            v = (v * pre_scale.astype(np.float16)[None, :]).astype(np.float16)      # __hmul2
        h = _fwht128_f32(v.astype(np.float32).reshape(rows, cols // 128, 128)).reshape(rows, cols)
        o = (h * r_scale).astype(np.float32).astype(np.float16)
        if post_scale is not None:
            o = (o * post_scale.astype(np.float16)[None, :]).astype(np.float16)
        return o
    assert x.dtype == np.float32
    v = x
    if pre_scale is not None:
        v = (v * pre_scale.astype(np.float32)[None, :]).astype(np.float32)
    h = _fwht128_f32(v.reshape(rows, cols // 128, 128)).reshape(rows, cols)
    h = (h * r_scale).astype(np.float32)
    if out_dtype == np.float16:
        h = h.astype(np.float16)
        if post_scale is not None:
            h = (h * post_scale.astype(np.float16)[None, :]).astype(np.float16)
        return h
    if post_scale is not None:
        h = (h * post_scale.astype(np.float32)[None, :]).astype(np.float32)
    return h


def hadamard_matrix_128() -> np.ndarray:
    """Sylvester H128 (+-1), symmetric.  util/hadamard.py:34-42."""
    h = np.array([[1.0]], dtype=np.float32)
    while h.shape[0] < 128:
        h = np.block([[h, h], [h, -h]])
    return h


# ------------------------------------------------------------------------------------------
# Quantized linear forward
# ------------------------------------------------------------------------------------------

def linear_forward(x: np.ndarray, trellis: np.ndarray, suh: np.ndarray, svh: np.ndarray, K: int, cb: int,
                   bias: np.ndarray | None = None, out_fp32: bool = False, w_hat: np.ndarray | None = None) -> np.ndarray:
    """y = ((x * suh) H) @ W_hat H * svh (+ bias) with the kernel path's rounding points
    (quant/exl3_gemm_inner.cuh:456-480, hadamard_inner.cuh:93-147,191-208,256-275,
    libtorch/linear.cpp:34-71): A_had rounded to fp16, fp32 accumulation, output Hadamard from fp32;
    fp16 out = round_fp16(had * r) *_fp16 svh;  fp32 out = all fp32."""
    m, k = x.shape
    assert x.dtype == np.float16
    if w_hat is None:
        w_hat = reconstruct(trellis, K, cb)
    xh = had_r_128(x, pre_scale=suh)                               # fp16
    acc = xh.astype(np.float32) @ w_hat.astype(np.float32)         # fp32 accumulate (order-free within tol)
    if out_fp32:
        y = had_r_128(acc.astype(np.float32), post_scale=svh)
        if bias is not None:
            y = (y + bias.astype(np.float32)[None, :]).astype(np.float32)
    else:
        y = had_r_128(acc.astype(np.float32), post_scale=svh, out_dtype=np.float16)
        if bias is not None:
            y = (y + bias.astype(np.float16)[None, :]).astype(np.float16)
    return y


def weight_tensor(trellis: np.ndarray, suh: np.ndarray, svh: np.ndarray, K: int, cb: int) -> np.ndarray:
    """Original-basis weights W = diag(suh) . H . W_hat . H . diag(svh), fp32 math (the parity target of
    reconstruct_had_slice at 2e-3 rel-to-max: tests/test_reconstruct_had.py:25-33;
    modules/quant/exl3.py:227-237 get_weight_tensor)."""
    w = reconstruct(trellis, K, cb).astype(np.float32)
    k, n = w.shape
    H = hadamard_matrix_128() * np.float32(1.0 / np.sqrt(128.0))
    w = np.einsum("ij,bjn->bin", H, w.reshape(k // 128, 128, n)).reshape(k, n)
    w = w * suh.astype(np.float32)[:, None]
    w = np.einsum("kbj,ji->kbi", w.reshape(k, n // 128, 128), H).reshape(k, n)
    w = w * svh.astype(np.float32)[None, :]
    return w.astype(np.float32)


def tp_slice(trellis, suh, svh, bias, first: int, last: int, dim: str):
    """TP shard of one quantized Linear.  modules/quant/exl3.py:284-330 (tp_import_split):
    out-split ('n'): trellis[:, first/16:last/16], svh[first:last], bias[first:last], full suh;
    in-split ('k'):  trellis[first/16:last/16], suh[first:last], full svh, bias only where first == 0."""
    assert first % 128 == 0 and last % 128 == 0
    if dim == "n":
        return (np.ascontiguousarray(trellis[:, first // 16:last // 16]), suh, svh[first:last],
                None if bias is None else bias[first:last])
    return (np.ascontiguousarray(trellis[first // 16:last // 16]), suh[first:last], svh,
            bias if (bias is not None and first == 0) else None)


# ------------------------------------------------------------------------------------------
# RMSNorm     exllamav3_ext/norm.cu:155-299 (kernel), modules/rmsnorm.py:65-79 (torch reference)
# ------------------------------------------------------------------------------------------

def rms_norm(x: np.ndarray, w: np.ndarray | None, eps: float, constant_bias: float = 0.0, constant_scale: float = 1.0,
             out_dtype=np.float16, residual_in: np.ndarray | None = None, add_residual_to: np.ndarray | None = None):
    """Kernel semantics: y = x * (w + bias) * rsqrt(mean(x^2) + eps) * scale, all fp32, one rounding to out dtype.
    residual_in (RES_IN, norm.cu:193-218): r += x rounded to r's dtype first, then y = norm(r); returns (y, r).
    add_residual_to (RES_POST, :257-266): y = existing + norm(x)."""
    xf = x.astype(np.float32)
    r_out = None
    if residual_in is not None:
        r_out = (residual_in.astype(np.float32) + xf).astype(residual_in.dtype)
        xf = r_out.astype(np.float32)
    dim = x.shape[-1]
    ss = (xf.astype(np.float64) ** 2).sum(-1, keepdims=True).astype(np.float32)
    rmf = (np.float32(1.0) / np.sqrt(ss / np.float32(dim) + np.float32(eps))).astype(np.float32) * np.float32(constant_scale)
    if w is not None:
        wf = w.astype(np.float32) + np.float32(constant_bias)
        y = xf * wf[None, :] * rmf
    else:
        y = xf * rmf
    if add_residual_to is not None:
        y = y + add_residual_to.astype(np.float32)
    y = y.astype(out_dtype)
    return (y, r_out) if residual_in is not None else y


# ------------------------------------------------------------------------------------------
# Attention output gates     exllamav3_ext/activation_kernels.cuh:132-139, 300-367; activation.cu:716-785
# ------------------------------------------------------------------------------------------

def _sigmoid_h(y: np.ndarray) -> np.ndarray:
    """activation_kernels.cuh:132-139: fp16 at every step -- e = fp16(exp(-y)), s = 1 +_h e, r = fp16(1 / s)."""
    y = np.asarray(y, dtype=np.float16)
    with np.errstate(over="ignore"):
        e = np.exp(-y.astype(np.float64)).astype(np.float32).astype(np.float16)
    sm = (np.float16(1.0) + e).astype(np.float16)
    with np.errstate(divide="ignore"):
        return (1.0 / sm.astype(np.float64)).astype(np.float32).astype(np.float16)


def mul_sigmoid(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """x * sigmoid(y), product in fp16 (activation_kernels.cuh:300-315); y has x's shape, or one gate per row of x's last dim (:318-336)."""
    x = np.asarray(x, dtype=np.float16)
    g = _sigmoid_h(y)
    if g.shape != x.shape:
        g = g[..., None]
    return (x * g).astype(np.float16)


def mul_softplus_broadcast(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """x [.., H, D] * softplus(y [.., H]): gate in fp32 = max(y, 0) + log1p(exp(-|y|)), one rounding of the product (activation_kernels.cuh:338-367)."""
    yf = np.asarray(y, dtype=np.float16).astype(np.float32)
    sp = (np.maximum(yf, np.float32(0)) + np.log1p(np.exp(-np.abs(yf)).astype(np.float32)).astype(np.float32)).astype(np.float32)
    return (np.asarray(x, dtype=np.float16).astype(np.float32) * sp[..., None]).astype(np.float16)


def add_sigmoid_gate(x: np.ndarray, y: np.ndarray, z: np.ndarray) -> np.ndarray:
    """z + x * sigmoid(y), fp32, one gate per row (activation_kernels.cuh:278-297; sigmoid = 1 / (1 + exp(-y)))."""
    g = (np.float32(1) / (np.float32(1) + np.exp(-np.asarray(y, np.float32)))).astype(np.float32)
    return (np.asarray(z, np.float32) + np.asarray(x, np.float32) * g).astype(np.float32)


def add_sigmoid_gate_proj(x: np.ndarray, y: np.ndarray, z: np.ndarray, w: np.ndarray) -> np.ndarray:
    """z + x * sigmoid(y . w) per row (activation_kernels.cuh:369-411): x, z fp32; y, w fp16; a gate below 1e-8 leaves the row untouched."""
    yw = (np.asarray(y, np.float16).astype(np.float32) @ np.asarray(w, np.float16).astype(np.float32).reshape(-1)).astype(np.float32)
    with np.errstate(over="ignore"):
        g = (np.float32(1) / (np.float32(1) + np.exp(-yw))).astype(np.float32)
    g = np.where(g < np.float32(1e-8), np.float32(0), g)
    return (np.asarray(z, np.float32) + np.asarray(x, np.float32) * g[:, None]).astype(np.float32)


def deinterleave_qg(qg: np.ndarray, head_dim: int):
    """[.., heads * 2 * head_dim] with (q head, g head) pairs per head -> (q, g) [.., heads * head_dim] (activation.cu:716-739)."""
    v = np.asarray(qg).reshape(qg.shape[:-1] + (-1, 2, head_dim))
    return (np.ascontiguousarray(v[..., 0, :]).reshape(qg.shape[:-1] + (-1,)), np.ascontiguousarray(v[..., 1, :]).reshape(qg.shape[:-1] + (-1,)))


# ------------------------------------------------------------------------------------------
# RoPE      exllamav3_ext/rope.cu:16-296, util/rope.py:102-140,365-432
# ------------------------------------------------------------------------------------------

ROPE_GPTJ, ROPE_NEOX = 1, 2      # util/rope.py RopeStyle


ROPE_NANOCHAT = 3


def _bf16_round(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest-even bfloat16, returned as float32 (norm weights of dtype bfloat16, rope.cu:221-231)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return u.view(np.float32)


def rope(q: np.ndarray, k: np.ndarray | None, inv_freq: np.ndarray, position: int = 0, positions=None, position_ids=None,
         rope_mode: int = ROPE_NEOX, attn_factor: float = 1.0, q_norm=None, k_norm=None, norm_eps: float = 1e-6,
         norm_constant_bias: float = 0.0, l4_beta: float = 0.0, l4_orig: int = 1, post_rope_norm: bool = False,
         rotate_dims: int = 1, rotate_offset: int = 0, norm_bf16: bool = False):
    """q (b, s, hq, d), k (b, s, hk, d) fp16 -> rotated fp16; the whole argument list of rope.cu:307-343.

    Position of token (b, t) = position + t, or positions[b] + t, or position_ids[b, t] (2-D) / position_ids[b, t, rdim] (3-D, one
    position per rotated sub-range: rope.cu:56-70).  inv_freq 1-D [p/2]: angle = inv_freq * pos; 2-D / 3-D: a table of angles indexed
    [batch (if 3-D)][pos][p/2] (rope.cu:80-87, host side :388-395).  p = 2 * inv_freq.shape[-1] is the rotated width ("partial_head_dim"):
    sub-range rdim covers elements [rotate_offset + p * rdim, + p); everything outside passes through (rope.cu:154-160).
    Stages, each leaving fp16 values like the kernel's shared half buffer (rope.cu:318-323):
      1. per-head RMSNorm when q_norm is given (k heads use k_norm): fp16 weights -> normalised value rounded to fp16, (w + bias) formed in fp16,
         product in fp16 (rope.cu:232-240); bf16 weights -> v * rmf * (float(w) + bias) in fp32, one rounding (rope.cu:221-231);
      2. rotation per sub-range, NEOX pairs (i, i + p/2), GPTJ pairs (2i, 2i+1), NANOCHAT = NEOX pairs with the opposite sign of sin (rope.cu:162-198);
      3. post_rope_norm: unweighted RMSNorm of the whole head (rope.cu:247-274);
      4. llama-4 scale on query heads only: 1 + beta * ln(1 + pos(rdim 0) // orig) on the whole head (rope.cu:90-94, :278-289, :297)."""
    inv_freq = np.asarray(inv_freq, dtype=np.float32)
    p2 = inv_freq.shape[-1]
    table = inv_freq.ndim > 1

    def one(x, norm_w, is_q):
        if x is None:
            return None
        b, s, h, d = x.shape
        assert rotate_dims >= 1 and rotate_offset >= 0 and rotate_offset + 2 * p2 * rotate_dims <= d
        xf = x.astype(np.float16).astype(np.float32)
        if q_norm is not None:
            ss = (xf.astype(np.float64) ** 2).sum(-1, keepdims=True).astype(np.float32)
            rmf = np.float32(1.0) / np.sqrt(ss / np.float32(d) + np.float32(norm_eps))
            if norm_bf16:
                w = _bf16_round(np.asarray(norm_w, dtype=np.float32)) + np.float32(norm_constant_bias)
                xf = ((xf * rmf).astype(np.float32) * w).astype(np.float16).astype(np.float32)
            else:
                vn = (xf * rmf).astype(np.float16)
                wh = (np.asarray(norm_w).astype(np.float16) + np.float16(norm_constant_bias)).astype(np.float16)
                xf = (vn * wh).astype(np.float16).astype(np.float32)

        def pos_of(rdim):
            if positions is not None:
                return (np.asarray(positions).reshape(b, 1) + np.arange(s)[None, :]).astype(np.int64)
            if position_ids is not None:
                pi = np.asarray(position_ids)
                return (pi.reshape(b, s, -1)[:, :, rdim if pi.ndim == 3 else 0]).astype(np.int64)
            return np.broadcast_to((position + np.arange(s))[None, :], (b, s)).astype(np.int64)

        out = xf.copy()
        for rdim in range(rotate_dims):
            pos = pos_of(rdim)
            if table:
                tb = inv_freq if inv_freq.ndim == 3 else np.broadcast_to(inv_freq[None], (b,) + inv_freq.shape)
                ang = tb[np.arange(b)[:, None], pos]                                               # (b, s, p/2)
            else:
                ang = pos[:, :, None].astype(np.float32) * inv_freq[None, None, :]
            sin = (np.sin(ang.astype(np.float64)) * attn_factor).astype(np.float32)[:, :, None, :]
            cos = (np.cos(ang.astype(np.float64)) * attn_factor).astype(np.float32)[:, :, None, :]
            o0 = rotate_offset + 2 * p2 * rdim
            seg = out[..., o0:o0 + 2 * p2]
            res = np.empty_like(seg)
            if rope_mode == ROPE_GPTJ:
                a, c = seg[..., 0::2], seg[..., 1::2]
                res[..., 0::2] = a * cos - c * sin
                res[..., 1::2] = c * cos + a * sin
            else:
                sg = np.float32(-1.0) if rope_mode == ROPE_NANOCHAT else np.float32(1.0)
                a, c = seg[..., :p2], seg[..., p2:]
                res[..., :p2] = a * cos - sg * c * sin
                res[..., p2:] = c * cos + sg * a * sin
            out[..., o0:o0 + 2 * p2] = res.astype(np.float16).astype(np.float32)
        if post_rope_norm:
            ss = (out.astype(np.float64) ** 2).sum(-1, keepdims=True).astype(np.float32)
            rmf = np.float32(1.0) / np.sqrt(ss / np.float32(d) + np.float32(norm_eps))
            out = (out * rmf).astype(np.float16).astype(np.float32)
        if l4_beta > 0.0 and is_q:
            sc = (np.float32(1.0) + np.float32(l4_beta) * np.log(np.float32(1.0) + (pos_of(0) // int(l4_orig)).astype(np.float32))).astype(np.float32)
            out = (out * sc[:, :, None, None]).astype(np.float16).astype(np.float32)
        return out.astype(np.float16)
    return one(q, q_norm, True), one(k, k_norm, False)


# ------------------------------------------------------------------------------------------
# KV-cache quantization     exllamav3_ext/cache/q_cache_kernels.cuh:29-236
# ------------------------------------------------------------------------------------------

def _fwht32_f32(v: np.ndarray) -> np.ndarray:
    v = v.astype(np.float32)
    idx = np.arange(32)
    for bit in (1, 2, 4, 8, 16):
        p = v[..., idx ^ bit]
        sgn = np.where(idx & bit, np.float32(-1), np.float32(1)).astype(np.float32)
        v = (v * sgn + p).astype(np.float32)
    return v


def kv_planes(bits: int):
    """Bit-plane widths, MSB plane first: set bits of `bits` among (8, 4, 2, 1).  q_cache_kernels.cuh:130-147."""
    return [w for w in (8, 4, 2, 1) if bits & w]


KV_RSQRT32 = np.float32(0.17677669529663689)


def _f32(x):
    return np.asarray(x, dtype=np.float32)


def _fma32(a, b, c):
    """fp32 fma of fp32 operands through float64 (a * b is exact there; the one extra rounding of the sum can differ from a true fma only when
    the float64 sum lands within 2^-29 ulp of an fp32 tie)."""
    return (_f32(a).astype(np.float64) * _f32(b).astype(np.float64) + _f32(c).astype(np.float64)).astype(np.float32)


class KvCompander:
    """Cubic compander of the KV level grid (reference: cache/lmq.cuh:56-85 LMCubic), fp32 in the reference's op order.
    decode: t = fma(2 q + 1, 2^-b, -1); level = t * fma(t^2, 1 - a, a).
    encode: q_half = x * inv_b * 0.5; delta = fma(q_half, q_half, p3^3); s = sqrt(delta); t = cbrt(q_half + s) + cbrt(q_half - s);
            q = clamp(floor(fma(t, 2^(b-1), 2^(b-1))), 0, 2^b - 1).   (np.cbrt / np.sqrt on float32 = libm cbrtf / sqrtf, as in oracle/_ref)"""
    def __init__(self, a: float):
        self.a = np.float32(a)
        self.b = np.float32(1.0) - self.a
        self.inv_b = np.float32(1.0) / self.b
        p3 = self.a * self.inv_b * (np.float32(1.0) / np.float32(3.0))
        self.p3_cub = np.float32(np.float32(p3 * p3) * p3)

    def decode(self, q: np.ndarray, bits: int) -> np.ndarray:
        t = _fma32(np.float32(2.0) * q.astype(np.float32) + np.float32(1.0), np.float32(1.0 / (1 << bits)), np.float32(-1.0))
        return (t * _fma32((t * t).astype(np.float32), self.b, self.a)).astype(np.float32)

    def encode(self, x: np.ndarray, bits: int) -> np.ndarray:
        q_half = ((_f32(x) * self.inv_b).astype(np.float32) * np.float32(0.5)).astype(np.float32)
        delta = _fma32(q_half, q_half, self.p3_cub)
        sq = np.sqrt(delta).astype(np.float32)
        t = (np.cbrt((q_half + sq).astype(np.float32)).astype(np.float32) + np.cbrt((q_half - sq).astype(np.float32)).astype(np.float32)).astype(np.float32)
        half = np.float32(1 << (bits - 1))
        return np.clip(np.floor(_fma32(t, half, half)), 0, (1 << bits) - 1).astype(np.uint32)


def kv_quant(x: np.ndarray, bits: int, compand_a: float = 0.0):
    """x (..., D) fp16, D % 32 == 0 -> (packed uint32 (..., D/32*bits), scales fp16 (..., D/32)).
    Per 32-group, in fp32 and in the reference's op order (q_cache_kernels.cuh:61-156):
      v = H32(x) * (1/sqrt(32));  s = max|v| + 1e-10;  inv_s = 1/s;
      q = clamp(floor(fma(v * inv_s, 2^(b-1), 2^(b-1))), 0, 2^b - 1);  stored scale = fp16(s);
    bit planes (8,4,2,1 widths present in b, MSB plane first), value e of the group at bit e*w of the plane.
    (The reference builds with --use_fast_math, so its 1/s is approximate; the oracle and the HIP kernel both use
    the IEEE division, which is what 'bit-exact vs the oracle' refers to.)"""
    shp = x.shape
    g = x.astype(np.float32).reshape(-1, 32)
    v = (_fwht32_f32(g) * KV_RSQRT32).astype(np.float32)
    s = (np.abs(v).max(-1, keepdims=True) + np.float32(1e-10)).astype(np.float32)
    inv_s = (np.float32(1.0) / s).astype(np.float32)
    half = np.float32(1 << (bits - 1))
    t = (v * inv_s).astype(np.float32)
    # fma(t, half, half): half is a power of two, so t*half is exact and the fma equals one rounded add
    qf = np.floor((t.astype(np.float64) * np.float64(half) + np.float64(half)).astype(np.float32))
    q = np.clip(qf, 0, (1 << bits) - 1).astype(np.uint32)         # (G, 32)
    if compand_a > 0.0:
        q = KvCompander(compand_a).encode(t, bits)                 # compand_a > 0: lmq.cuh encoder instead of the midpoint grid (:111-119)
    words = []
    rem = bits
    e = np.arange(32)
    for w in kv_planes(bits):
        rem -= w
        pv = (q >> np.uint32(rem)) & np.uint32((1 << w) - 1)       # (G, 32)
        widx = (e * w) // 32
        shift = ((e * w) % 32).astype(np.uint64)
        plane = np.zeros((q.shape[0], w), dtype=np.uint32)
        for wi in range(w):
            sel = widx == wi
            plane[:, wi] = (pv[:, sel].astype(np.uint64) << shift[sel][None, :]).sum(-1, dtype=np.uint64).astype(np.uint32)
        words.append(plane)
    packed = np.concatenate(words, axis=-1)                        # (G, bits)
    D = shp[-1]
    return packed.reshape(shp[:-1] + (D // 32 * bits,)), s.astype(np.float16).reshape(shp[:-1] + (D // 32,))


def kv_dequant(packed: np.ndarray, scales: np.ndarray, bits: int, compand_a: float = 0.0) -> np.ndarray:
    """Inverse of kv_quant (q_cache_kernels.cuh:160-236): s' = fp32(scale) * (1/sqrt(32)); sm = s' * 2^-(b-1);
    v = (q - (2^(b-1) - 0.5)) * sm; x = H32(v) -> fp16."""
    shp = scales.shape
    pk = packed.reshape(-1, bits).astype(np.uint32)
    q = np.zeros((pk.shape[0], 32), dtype=np.uint32)
    off = 0
    e = np.arange(32)
    for w in kv_planes(bits):
        widx = (e * w) // 32
        shift = ((e * w) % 32).astype(np.uint32)
        pv = (pk[:, off + widx] >> shift[None, :]) & np.uint32((1 << w) - 1)
        q = (q << np.uint32(w)) | pv
        off += w
    s = (scales.reshape(-1, 1).astype(np.float32) * KV_RSQRT32).astype(np.float32)
    half = np.float32(1 << (bits - 1))
    sm = (s * (np.float32(1.0) / half)).astype(np.float32)
    v = ((q.astype(np.float32) - (half - np.float32(0.5))) * sm).astype(np.float32)
    if compand_a > 0.0:
        v = (KvCompander(compand_a).decode(q, bits) * s).astype(np.float32)      # q_cache_kernels.cuh:205-212: level * (scale / sqrt(32))
    x = _fwht32_f32(v)
    return x.astype(np.float16).reshape(shp[:-1] + (shp[-1] * 32,))


def kv_quant_paged(k_in, k_out, k_scales, v_in, v_out, v_scales, cache_seqlens, block_table, k_bits: int, v_bits: int,
                   page_size: int = 256, in_contiguous: bool = True, seq_len: int | None = None, compand_a: float = 0.0):
    """quant_cache_paged_kernel's addressing (cache/q_cache_kernels.cuh:289-325): new token j of sequence b lands at logical position
    cache_seqlens[b] + j, i.e. physical row block_table[b][pos // page] * page + pos % page of the flat (pages * page) cache.
    k_in / v_in: (bsz, seq_len, D) fp16 when in_contiguous, else the flat (pages * page, D) staging cache read at the same physical row.
    k_out / v_out: uint32 (pages, page, D/32*bits), scales fp16 (pages, page, D/32): updated in place."""
    bsz = block_table.shape[0]
    if in_contiguous: seq_len = k_in.shape[1]
    D = k_in.shape[-1]
    ko, ks = k_out.reshape(-1, D // 32 * k_bits), k_scales.reshape(-1, D // 32)
    vo, vs = v_out.reshape(-1, D // 32 * v_bits), v_scales.reshape(-1, D // 32)
    kf, vf = (k_in, v_in) if in_contiguous else (k_in.reshape(-1, D), v_in.reshape(-1, D))
    for b in range(bsz):
        for j in range(seq_len):
            pos = int(cache_seqlens[b]) + j
            row = int(block_table[b, pos // page_size]) * page_size + pos % page_size
            kx = kf[b, j] if in_contiguous else kf[row]
            vx = vf[b, j] if in_contiguous else vf[row]
            pk, sc = kv_quant(kx[None, :], k_bits, compand_a); ko[row] = pk[0]; ks[row] = sc[0]
            pk, sc = kv_quant(vx[None, :], v_bits, compand_a); vo[row] = pk[0]; vs[row] = sc[0]


def kv_dequant_paged(k_in, k_scales, k_out, v_in, v_scales, v_out, cache_seqlens, block_table, k_bits: int, v_bits: int,
                     page_size: int = 256, bonus_len: int = 0, sliding_window: int = 0, compand_a: float = 0.0, compact_out: bool = False):
    """dequant_cache_paged_kernel (cache/q_cache_kernels.cuh:342-399) as launched by cache/q_cache.cu:271-346 (dequant_cache_paged:
    compact_out False, bonus_len 0) and :358-433 (dequant_cache_paged_window: sliding_window <= 0, compact_out True).
    Logical positions [0, cache_seqlens[b] + bonus_len) of every sequence are dequantized; the other rows stay untouched.  With
    sliding_window > 0 the kernel also skips whole thread blocks that end at or before max_len - window.  A block is 8 warps x 32 iterations =
    256 consecutive "chunks" of the sequence (a chunk = 4 groups of one token, cpt = ceil(groups_per_token / 4) chunks per token), so chunk c
    of token t is skipped iff floor(256 * (floor((t * cpt + c) / 256) + 1) / cpt) <= max_len - window -- groups of one row can be split
    when cpt does not divide 256.  compact_out: page p of sequence b is written at rows (b * pages_per_seq + p) * page_size + ... of the
    output instead of the physical page."""
    D = k_out.shape[-1]
    gpt = D // 32
    cpt = (gpt + 3) // 4
    pps = block_table.shape[1]
    ki, ks = k_in.reshape(-1, gpt * k_bits), k_scales.reshape(-1, gpt)
    vi, vs = v_in.reshape(-1, gpt * v_bits), v_scales.reshape(-1, gpt)
    ko, vo = k_out.reshape(-1, D), v_out.reshape(-1, D)
    for b in range(block_table.shape[0]):
        max_len = int(cache_seqlens[b]) + bonus_len
        for pos in range(max_len):
            row = int(block_table[b, pos // page_size]) * page_size + pos % page_size
            orow = (b * pps + pos // page_size) * page_size + pos % page_size if compact_out else row
            kd = kv_dequant(ki[row][None, :], ks[row][None, :], k_bits, compand_a)[0]
            vd = kv_dequant(vi[row][None, :], vs[row][None, :], v_bits, compand_a)[0]
            for c in range(cpt):
                if sliding_window > 0 and (256 * ((pos * cpt + c) // 256 + 1)) // cpt <= max_len - sliding_window:
                    continue
                sl = slice(c * 128, min((c + 1) * 128, D))
                ko[orow, sl] = kd[sl]; vo[orow, sl] = vd[sl]


def act_mul(g: np.ndarray, u: np.ndarray, act: str, act_limit: float = 0.0) -> np.ndarray:
    """z = act(g) * u of the reference's gated MLP (activation_kernels.cuh:142-254 act_mul_kernel_h / _f; activations :9-128).  act: "silu", "gelu",
    "relu2", "relu", "silu_oai".  fp16 inputs follow act_mul_kernel_h: the activation result is rounded to fp16 (gelu / relu2 are computed in fp32 and
    rounded, :40-49, :67-73), the clamps and the product are fp16 operations; fp32 inputs follow act_mul_kernel_f (fp32 throughout, one rounding, clamp
    to the finite fp16 range).  silu_oai (:118-128) is fp32 in both."""
    half_in = g.dtype == np.float16
    gf, uf = g.astype(np.float32), u.astype(np.float32)
    L = np.float32(act_limit)
    if act == "silu_oai":
        if act_limit != 0.0:
            gf = np.minimum(gf, L); uf = np.minimum(np.maximum(uf, -L), L)
        r = (uf + np.float32(1)) * (gf / (np.float32(1) + np.exp(np.float32(-1.702) * gf)))
        r = r.astype(np.float32)
        return (r if half_in else np.clip(r, -65504.0, 65504.0)).astype(np.float16)
    if act == "silu":
        if half_in:
            # _silu(half2): every step rounded to fp16 (:21-31)
            with np.errstate(over="ignore"):                 # exp(-g) beyond the fp16 range is inf there too: 1 / inf = 0
                e = np.exp(-g.astype(np.float32)).astype(np.float16)
                sm = (np.float16(1) + e).astype(np.float16)
            rc = (np.float32(1) / sm.astype(np.float32)).astype(np.float16)
            a = (g * rc).astype(np.float16)
        else:
            a = gf / (np.float32(1) + np.exp(-gf))
    elif act == "gelu":
        # tanh_opt of compat.cuh:12-18, the exp-based branch the reference builds on AMD: copysign((1 - e) / (e + 1), x), e = exp(-2 |x|)
        targ = (np.float32(0.797884560803) * (gf + np.float32(0.044715) * gf * gf * gf)).astype(np.float32)
        e = np.exp(np.float32(-1) * np.abs(np.float32(2) * targ)).astype(np.float32)
        th = np.copysign(((np.float32(1) - e) / (e + np.float32(1))).astype(np.float32), targ)
        a = (np.float32(0.5) * gf * (np.float32(1) + th)).astype(np.float32)
    elif act == "relu2":
        a = np.maximum(gf, np.float32(0)) ** 2
    elif act == "relu":
        a = np.maximum(gf, np.float32(0))
    else:
        raise ValueError(act)
    if half_in:
        a = a.astype(np.float16); uu = u
        if act_limit != 0.0:
            Lh = np.float16(act_limit)
            uu = np.minimum(np.maximum(uu, -Lh), Lh); a = np.minimum(a, Lh)
        return (a.astype(np.float32) * uu.astype(np.float32)).astype(np.float16)
    a = a.astype(np.float32)
    if act_limit != 0.0:
        uf = np.minimum(np.maximum(uf, -L), L); a = np.minimum(a, L)
    return np.clip(a * uf, -65504.0, 65504.0).astype(np.float16)


def softcap(x: np.ndarray, scale: float) -> np.ndarray:
    """y = scale * tanh(x / scale) in fp32, rounded to x's dtype (exllamav3_ext/softcap.cu:11-52)."""
    v = x.astype(np.float32) / np.float32(scale)
    return (np.tanh(v).astype(np.float32) * np.float32(scale)).astype(x.dtype)


def linear_wrapper_forward(x: np.ndarray, trellis, suh, svh, K: int, cb: int, out_features_unpadded: int, trim: bool = True,
                           softcap_scale: float = 0.0, post_scale: float = 1.0, out_fp32: bool = False) -> np.ndarray:
    """Linear.forward around the quantized inner op (modules/linear.py:561-602): zero-extend x to the padded in_features, inner forward, trim the
    padded output columns, softcap, post_scale."""
    k = suh.shape[0]
    if x.shape[-1] < k:
        x = np.concatenate([x, np.zeros(x.shape[:-1] + (k - x.shape[-1],), x.dtype)], axis=-1)
    y = linear_forward(x, trellis, suh, svh, K, cb, out_fp32=out_fp32)
    if trim and out_features_unpadded != y.shape[-1]:
        y = np.ascontiguousarray(y[..., :out_features_unpadded])
    if softcap_scale != 0.0:
        y = softcap(y, softcap_scale)
    if post_scale != 1.0:
        y = (y.astype(np.float32) * np.float32(post_scale)).astype(y.dtype) if y.dtype == np.float16 else (y * np.float32(post_scale)).astype(np.float32)
    return y


# ------------------------------------------------------------------------------------------
# Synthetic tensors
# ------------------------------------------------------------------------------------------

def synth_linear(k: int, n: int, K: int, seed: int | None = None, realistic: bool = False):
    """trellis = randint(0, 65536, (k/16, n/16, 16K)) -> int16 (tests/test_reconstruct_had.py:44-45);
    suh/svh = sign(randn) fp16, or the log-normal 'realistic' variant of SURVEY 8(d)."""
    rng = np.random.default_rng(k * 7 + n + K if seed is None else seed)
    trellis = rng.integers(0, 65536, size=(k // 16, n // 16, 16 * K), dtype=np.uint16).view(np.int16)
    su = np.where(rng.standard_normal(k) < 0, -1.0, 1.0)
    sv = np.where(rng.standard_normal(n) < 0, -1.0, 1.0)
    if realistic:
        su = su * 0.02 * np.exp(0.2 * rng.standard_normal(k))
        sv = sv * np.exp(0.2 * rng.standard_normal(n))
    return trellis, su.astype(np.float16), sv.astype(np.float16)


def routing_std(hidden: np.ndarray, gate: np.ndarray, k: int, bias: np.ndarray | None = None, per_expert_scale: np.ndarray | None = None):
    """MoE router (exllamav3_ext/routing.cu:457-590 + routing_gemv): scores = fp16(hidden @ gate) ; top-k logits (descending, ties to the lower
    index) ; weights = fp16(softmax over the k selected logits [x per_expert_scale[expert], fp32 values of a bf16 tensor, routing.cu:587-588]).
    Returns (scores fp16 [b, E], indices int64 [b, k], weights fp16 [b, k])."""
    scores = (hidden.astype(np.float32) @ gate.astype(np.float32)).astype(np.float16)
    logits = scores.astype(np.float32) + (bias.astype(np.float32) if bias is not None else 0.0)
    order = np.argsort(-logits, axis=-1, kind="stable")[:, :k]
    sel = np.take_along_axis(logits, order, axis=-1)
    e = np.exp(sel - sel[:, :1])
    wf = (e / (e.sum(-1, keepdims=True) + 1e-20)).astype(np.float32)
    if per_expert_scale is not None:
        wf = (wf * per_expert_scale.astype(np.float32)[order]).astype(np.float32)
    return scores, order.astype(np.int64), wf.astype(np.float16)


def attn_prefill(q: np.ndarray, k: np.ndarray, v: np.ndarray, kv_lens, scale: float | None = None) -> np.ndarray:
    """Causal attention of a chunk of new tokens over the (dequantized) cache, as the reference's prefill path attends
    (flash_attn_with_kvcache(..., causal=True) after cache/quant.py:83-117): q (b, T, hq, d) fp16; k / v (b, S, hkv, d) fp16 with kv_lens[b] valid
    rows INCLUDING the T new tokens; query i sees keys 0 .. kv_lens[b] - T + i.  fp32 math, fp16 result.
    flash-attn 2 is an optional, un-vendored dependency of the reference (no version pin in requirements.txt; call site
    modules/attention_fn/flash_attn_2.py:12-38); pinned against the reference's own pure-torch equivalent instead
    (modules/attention_fn/torch.py:81-160, fixtures attn_* of tests/golden/ref_python.npz, tests/test_oracle_pins.py)."""
    b, T, hq, d = q.shape
    hkv = k.shape[2]
    gq = hq // hkv
    sc = np.float32(scale if scale is not None else d ** -0.5)
    out = np.zeros((b, T, hq, d), dtype=np.float32)
    for bi in range(b):
        L = int(kv_lens[bi]); ctx = L - T
        mask = np.arange(L)[None, :] <= (ctx + np.arange(T))[:, None]
        for h in range(hq):
            kk = k[bi, :L, h // gq].astype(np.float32); vv = v[bi, :L, h // gq].astype(np.float32)
            s = (q[bi, :, h].astype(np.float32) @ kk.T) * sc
            s = np.where(mask, s, np.float32(-1e30))
            p = np.exp(s - s.max(-1, keepdims=True)); p = np.where(mask, p, 0.0); p /= p.sum(-1, keepdims=True)
            out[bi, :, h] = p @ vv
    return out.astype(np.float16)


def attn_decode_qcache(q: np.ndarray, k_deq: np.ndarray, v_deq: np.ndarray, lens, scale: float | None = None, sinks=None) -> np.ndarray:
    """Decode attention of one new token per sequence over the DEQUANTIZED cache (what the reference attends to after dequant_cache_paged,
    libtorch/attention.cpp:246-504): q (b, hq, d) fp16; k_deq / v_deq (b, T, hkv, d) fp16 = kv_dequant of the cache; lens[b] tokens valid.
    GQA: q head h uses kv head h // (hq / hkv).  fp32 softmax / accumulation, fp16 result.
    sinks (hq,) fp32: learned per-head sink logits in the units of the scaled scores, in the softmax denominator only -- the combine kernel's
    HAS_SINKS branch (modules/attention_fn/triton_paged.py:1030-1050); the reference's own torch restatement: concatenate the logit, softmax, drop
    the column (tests/test_dsa_kernels.py:54-62)."""
    b, hq, d = q.shape
    hkv = k_deq.shape[2]
    gq = hq // hkv
    sc = np.float32(scale if scale is not None else d ** -0.5)
    out = np.zeros((b, hq, d), dtype=np.float32)
    for bi in range(b):
        L = int(lens[bi])
        for h in range(hq):
            kk = k_deq[bi, :L, h // gq].astype(np.float32); vv = v_deq[bi, :L, h // gq].astype(np.float32)
            s = (kk @ q[bi, h].astype(np.float32)) * sc
            if sinks is not None:
                # learned sink logit of head h: in the denominator only (modules/attention_fn/triton_paged.py:1030-1050)
                m = max(np.float32(s.max()) if L else np.float32(-np.inf), np.float32(sinks[h]))
                p = np.exp(s - m)
                out[bi, h] = (p @ vv) / (p.sum() + np.exp(np.float32(sinks[h]) - m))
                continue
            p = np.exp(s - s.max()); p /= p.sum()
            out[bi, h] = p @ vv
    return out.astype(np.float16)
