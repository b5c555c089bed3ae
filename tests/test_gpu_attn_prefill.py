"""Prefill (multi-token) causal attention over paged fp16 K/V (exl3_attn_prefill.hip) against the oracle: the reference's prefill path attends with
flash_attn_with_kvcache(causal=True) over the pages dequant_cache_paged expands (cache/quant.py:83-117).  Tolerance: fp16 probabilities into the
second product and an fp16 result -> 2e-2 of the output RMS (the decode-attention tests use 1e-2 with fp32 probabilities)."""
import os, sys
import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import exl3_oracle as o

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("hd,hq,hkv", [(128, 8, 2), (64, 8, 4), (128, 4, 4)])
@pytest.mark.parametrize("q_len,ctx", [(64, [0, 0]), (100, [300, 17]), (1, [255, 256]), (257, [5, 700])])
@pytest.mark.parametrize("min_wgs", ["0", "1000000"])
def test_attn_prefill_paged_matches_oracle(dev, monkeypatch, hd, hq, hkv, q_len, ctx, min_wgs):
    """Chunks that start an empty cache, continue a context (different lengths per sequence, page-crossing, a page-aligned one), a single
    token (== decode semantics) and a chunk that is not a multiple of the 64-query workgroup tile; permuted block table; GQA 4 / 2 / 1."""
    from exllamav3_amd import ext
    monkeypatch.setenv("EXL3_HIP_ATTN_PREFILL_W64_MIN_WGS", min_wgs)         # the round-4 kernel wherever it applies / the round-3 kernel everywhere
    page, bsz = 256, 2
    rng = np.random.default_rng(hd + q_len)
    kv_lens = np.array([c + q_len for c in ctx], np.int32)
    pps = int((kv_lens.max() + page - 1) // page) + 1
    npages = bsz * pps + 1
    bt = rng.permutation(npages)[: bsz * pps].reshape(bsz, pps).astype(np.int32)
    k = rng.standard_normal((bsz, pps * page, hkv, hd)).astype(np.float16); v = rng.standard_normal((bsz, pps * page, hkv, hd)).astype(np.float16)
    q = (rng.standard_normal((bsz, q_len, hq, hd)) * 1.5).astype(np.float16)
    kp = np.full((npages, page, hkv, hd), np.nan, np.float16); vp = kp.copy()      # rows beyond the length / unmapped pages must not matter
    for b in range(bsz):
        for pg in range(pps):
            kp[bt[b, pg]] = k[b, pg * page:(pg + 1) * page]; vp[bt[b, pg]] = v[b, pg * page:(pg + 1) * page]
        kp[bt[b, kv_lens[b] // page], kv_lens[b] % page:] = np.nan if kv_lens[b] % page else kp[bt[b, kv_lens[b] // page], 0:]
    out = torch.full((bsz, q_len, hq, hd), float("nan"), dtype=torch.half, device=dev)
    ext.attn_prefill_paged(_t(q, dev), out, _t(kp, dev), _t(vp, dev), _t(bt, dev), _t(kv_lens, dev))
    ref = o.attn_prefill(q, k, v, kv_lens).astype(np.float32)
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() / np.sqrt((ref ** 2).mean()) < 2e-2


def test_attn_prefill_single_token_equals_decode_attention(dev):
    """q_len == 1 over a quantized cache: dequant_cache_paged + attn_prefill_paged (the reference's route) against attn_decode_qcache (this
    build's quant-cache-direct decode kernel): the two attention implementations agree to the quantization-independent tolerance."""
    from exllamav3_amd import ext
    hd, hq, hkv, page, bsz, pps, bits = 128, 8, 2, 256, 2, 2, 4
    rng = np.random.default_rng(1)
    lens = np.array([301, 77], np.int32)
    G = hkv * hd // 32
    npages = bsz * pps
    bt = rng.permutation(npages).reshape(bsz, pps).astype(np.int32)
    ck = rng.standard_normal((npages, page, hkv * hd)).astype(np.float16); cv = rng.standard_normal((npages, page, hkv * hd)).astype(np.float16)
    kq, ks = o.kv_quant(ck, bits); vq, vs = o.kv_quant(cv, bits)
    T = lambda a: _t(a.view(np.int32) if a.dtype == np.uint32 else a, dev)
    dkq, dks, dvq, dvs = T(kq), T(ks), T(vq), T(vs)
    q = rng.standard_normal((bsz, hq, hd)).astype(np.float16)
    o_dec = torch.empty((bsz, hq, hd), dtype=torch.half, device=dev)
    ext.attn_decode_qcache(_t(q, dev), o_dec, dkq, dks, dvq, dvs, _t(bt, dev), _t(lens, dev), pps * page)
    kd = torch.zeros((npages, page, hkv * hd), dtype=torch.half, device=dev); vd = torch.zeros_like(kd)
    ext.dequant_cache_paged(dkq, dks, kd, dvq, dvs, vd, _t(lens, dev), _t(bt, dev), page)
    o_pre = torch.empty((bsz, 1, hq, hd), dtype=torch.half, device=dev)
    ext.attn_prefill_paged(_t(q, dev).view(bsz, 1, hq, hd), o_pre, kd.view(npages, page, hkv, hd), vd.view(npages, page, hkv, hd), _t(bt, dev), _t(lens, dev))
    a, b_ = o_dec.float().cpu().numpy(), o_pre.float().cpu().numpy().reshape(bsz, hq, hd)
    assert np.abs(a - b_).max() / np.sqrt((a ** 2).mean()) < 2e-2


def test_prefill_chunk_with_attention_matches_oracle(dev):
    """prefill_chunk with the attention core switched on (fused q|k|v GEMM -> RoPE -> quantized append -> dequant_cache_paged -> causal attention
    over the pages -> o_proj -> MLP) through one small layer against the oracle composition on EVERY token row (attention couples the rows):
    residual stream after the layer and the last token's logits."""
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    torch.manual_seed(7)                                      # the chunk's input comes from the global generator: same rows whatever ran before
    shape = LlamaShape("tiny", 256, 512, 1, 4, 2, 128, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4)
    model.prefill_attention = True
    T = 300
    logits = model.prefill_chunk(T).float().cpu().numpy()
    assert np.isfinite(logits).all()
    _np = lambda t: t.detach().cpu().numpy()
    lin = lambda l, a, **kw: o.linear_forward(a, _np(l.trellis), _np(l.suh), _np(l.svh), l.K, 1 if l.mcg else (2 if l.mul1 else 0), **kw)
    x = _np(model.px0)
    L = model.layers[0]
    xn = o.rms_norm(x, _np(L["norm1"]), model.eps)
    q, k, v = lin(L["q"], xn), lin(L["k"], xn), lin(L["v"], xn)
    q4, k4 = o.rope(q.reshape(1, T, model.hq, 128), k.reshape(1, T, model.hkv, 128), _np(model.inv_freq), position=0, rope_mode=o.ROPE_NEOX)
    kq, ks = o.kv_quant(k4.reshape(T, -1), 4); vq, vs = o.kv_quant(v.reshape(T, -1), 4)
    kd = o.kv_dequant(kq, ks, 4).reshape(1, T, model.hkv, 128); vd = o.kv_dequant(vq, vs, 4).reshape(1, T, model.hkv, 128)
    ao = o.attn_prefill(q4, kd, vd, np.array([T]))
    ov = lin(L["o"], ao.reshape(T, -1), out_fp32=True)
    xn2, x1 = o.rms_norm(ov, _np(L["norm2"]), model.eps, residual_in=x)
    gf, uf = lin(L["gate"], xn2).astype(np.float32), lin(L["up"], xn2).astype(np.float32)
    a = (gf / (1 + np.exp(-gf)) * uf).astype(np.float16)
    d = lin(L["down"], a, out_fp32=True)
    x2 = (x1.astype(np.float32) + d).astype(np.float16)
    got_x = _np(model.px_out).astype(np.float32)
    rel = lambda g, r: np.abs(g - r.astype(np.float32)).max() / np.sqrt((r.astype(np.float32) ** 2).mean())
    assert rel(got_x, x2) < 3e-2
    xl = o.rms_norm(x2[-1:], _np(model.final_norm), model.eps)
    ref = lin(model.lm_head, xl).astype(np.float32)
    assert rel(logits, ref) < 5e-2


def test_attn_prefill_rows_without_keys_are_zero(dev):
    """cache_seqlens[b] < q_len is outside the contract (the chunk is appended before the call), but the rows that then see no key must come out as
    zeros and the others as the causal result, never as garbage; bsz == 0 is a no-op."""
    from exllamav3_amd import ext
    hd, hq, hkv, page, q_len = 128, 4, 2, 256, 70
    rng = np.random.default_rng(3)
    kv_len = 50                                                              # queries 0 .. 19 sit at positions -20 .. -1
    k = rng.standard_normal((1, page, hkv, hd)).astype(np.float16); v = rng.standard_normal((1, page, hkv, hd)).astype(np.float16)
    q = rng.standard_normal((1, q_len, hq, hd)).astype(np.float16)
    out = torch.full((1, q_len, hq, hd), float("nan"), dtype=torch.half, device=dev)
    bt = torch.zeros((1, 1), dtype=torch.int32, device=dev)
    ext.attn_prefill_paged(_t(q, dev), out, _t(k, dev), _t(v, dev), bt, torch.tensor([kv_len], dtype=torch.int32, device=dev))
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all() and not got[0, :q_len - kv_len].any()
    ref = o.attn_prefill(q[:, q_len - kv_len:], k, v, np.array([kv_len])).astype(np.float32)
    assert np.abs(got[:, q_len - kv_len:] - ref).max() / np.sqrt((ref ** 2).mean()) < 2e-2
    empty = torch.empty((0, q_len, hq, hd), dtype=torch.half, device=dev)
    ext.attn_prefill_paged(empty, torch.empty_like(empty), _t(k, dev), _t(v, dev), torch.zeros((0, 1), dtype=torch.int32, device=dev),
                           torch.zeros((0,), dtype=torch.int32, device=dev))


@pytest.fixture
def matrix_kernel(monkeypatch):
    """The round-4 kernel is dispatched from ~192 workgroups on (the older one fills the chip better below); these tests run it on small shapes."""
    monkeypatch.setenv("EXL3_HIP_ATTN_PREFILL_W64_MIN_WGS", "0")


@pytest.mark.parametrize("hd,hq,hkv", [(128, 8, 2), (128, 8, 1), (128, 16, 4), (128, 6, 2), (128, 12, 2), (128, 7, 1), (128, 14, 2), (128, 10, 2),
                                       (64, 8, 2), (64, 32, 8), (64, 6, 2), (64, 7, 1)])
@pytest.mark.parametrize("q_len,ctx", [(64, [0, 63]), (65, [1, 256]), (200, [700, 0]), (512, [37, 1000])])
def test_attn_prefill_matrix_kernel_shapes(dev, matrix_kernel, hd, hq, hkv, q_len, ctx):
    """The round-4 kernel (head_dim 128 and 64, chunks of >= 64 tokens: 8 waves x 32 queries, 32x32x16 matrix instructions, scores started at minus the
    running maximum, row sums on the matrix pipe): group sizes 4 and 8, and 3 / 5 / 6 / 7 (the last group of four head slots of a kv head partly
    empty: those waves duplicate a head and store nothing), two sequences with different contexts (page-aligned, page-crossing, empty), chunk
    lengths on / off the 64-query tile, permuted block table, NaN in every unmapped row."""
    from exllamav3_amd import ext
    page, bsz = 256, 2
    rng = np.random.default_rng(hq * 1000 + q_len + hd)
    kv_lens = np.array([c + q_len for c in ctx], np.int32)
    pps = int((kv_lens.max() + page - 1) // page) + 1
    npages = bsz * pps + 1
    bt = rng.permutation(npages)[: bsz * pps].reshape(bsz, pps).astype(np.int32)
    k = rng.standard_normal((bsz, pps * page, hkv, hd)).astype(np.float16); v = rng.standard_normal((bsz, pps * page, hkv, hd)).astype(np.float16)
    q = (rng.standard_normal((bsz, q_len, hq, hd)) * 1.5).astype(np.float16)
    kp = np.full((npages, page, hkv, hd), np.nan, np.float16); vp = kp.copy()
    for b in range(bsz):
        for pg in range(pps):
            kp[bt[b, pg]] = k[b, pg * page:(pg + 1) * page]; vp[bt[b, pg]] = v[b, pg * page:(pg + 1) * page]
        if kv_lens[b] % page:
            kp[bt[b, kv_lens[b] // page], kv_lens[b] % page:] = np.nan; vp[bt[b, kv_lens[b] // page], kv_lens[b] % page:] = np.nan
    out = torch.full((bsz, q_len, hq, hd), float("nan"), dtype=torch.half, device=dev)
    ext.attn_prefill_paged(_t(q, dev), out, _t(kp, dev), _t(vp, dev), _t(bt, dev), _t(kv_lens, dev))
    ref = o.attn_prefill(q, k, v, kv_lens).astype(np.float32)
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() / np.sqrt((ref ** 2).mean()) < 2e-2


@pytest.mark.parametrize("hd", [128, 64])
def test_attn_prefill_matrix_kernel_reference_maximum_moves_late(dev, matrix_kernel, hd):
    """The running maximum of the round-4 kernel is a REFERENCE that moves only when a score exceeds it by 2^8 (rarely taken branches adjust the tile's
    scores, the chains' start value and, after the tile's P V, the accumulators).  Random keys of one scale never take those branches after the first
    tile; here the key norms grow by 10x along the sequence and single keys are aligned with the queries, so the maximum of every query jumps by far
    more than 8 log2 units several times, in different tiles for different queries."""
    from exllamav3_amd import ext
    hq, hkv, page, q_len, ctx = 8, 2, 256, 320, 448
    rng = np.random.default_rng(11 + hd)
    L = ctx + q_len
    pps = (L + page - 1) // page
    k = rng.standard_normal((1, pps * page, hkv, hd)).astype(np.float32)
    v = rng.standard_normal((1, pps * page, hkv, hd)).astype(np.float16)
    q = (rng.standard_normal((1, q_len, hq, hd)) * 1.5).astype(np.float16)
    k *= np.geomspace(0.3, 3.0, pps * page)[None, :, None, None]                      # norms grow along the sequence
    for pos in (130, 470, 600, 700):                                                    # spikes: keys aligned with the mean query of their kv head
        for h in range(hkv):
            k[0, pos, h] = 4.0 * q[0, :, h * (hq // hkv)].astype(np.float32).mean(0) + k[0, pos, h]
    k = k.astype(np.float16)
    kv_lens = np.array([L], np.int32)
    bt = np.arange(pps, dtype=np.int32)[None, :]
    out = torch.full((1, q_len, hq, hd), float("nan"), dtype=torch.half, device=dev)
    ext.attn_prefill_paged(_t(q, dev), out, _t(k.reshape(pps, page, hkv, hd), dev), _t(v.reshape(pps, page, hkv, hd), dev), _t(bt, dev), _t(kv_lens, dev))
    ref = o.attn_prefill(q, k, v, kv_lens).astype(np.float32)
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() / np.sqrt((ref ** 2).mean()) < 2e-2


def test_attn_prefill_matrix_kernel_rows_without_keys_are_zero(dev, matrix_kernel):
    """cache_seqlens[b] < q_len on the round-4 kernel (four query heads per kv head): rows that see no key are zeros, the others the causal result."""
    from exllamav3_amd import ext
    hd, hq, hkv, page, q_len = 128, 8, 2, 256, 150
    rng = np.random.default_rng(5)
    kv_len = 100                                                             # queries 0 .. 49 sit at negative positions
    k = rng.standard_normal((1, page, hkv, hd)).astype(np.float16); v = rng.standard_normal((1, page, hkv, hd)).astype(np.float16)
    q = rng.standard_normal((1, q_len, hq, hd)).astype(np.float16)
    out = torch.full((1, q_len, hq, hd), float("nan"), dtype=torch.half, device=dev)
    bt = torch.zeros((1, 1), dtype=torch.int32, device=dev)
    ext.attn_prefill_paged(_t(q, dev), out, _t(k, dev), _t(v, dev), bt, torch.tensor([kv_len], dtype=torch.int32, device=dev))
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all() and not got[0, :q_len - kv_len].any()
    ref = o.attn_prefill(q[:, q_len - kv_len:], k, v, np.array([kv_len])).astype(np.float32)
    assert np.abs(got[:, q_len - kv_len:] - ref).max() / np.sqrt((ref ** 2).mean()) < 2e-2
