"""GPU parity of RoPE and the KV-cache quantization kernels (through the C-ABI)."""
import numpy as np
import pytest
import torch
from oracle import exl3_oracle as o

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("style,mode", [("neox", 2), ("gptj", 1)])
def test_rope_vs_reference_python_fixtures(dev, golden, style, mode):
    # fixtures produced by the reference's RoPE.apply_torch (tests/test_rope.py tolerance 3e-3)
    from exllamav3_amd import ext
    q, k, inv = golden[f"rope_{style}_q"], golden[f"rope_{style}_k"], golden[f"rope_{style}_inv_freq"]
    cases = [("pos37", dict(position=37, positions=None, position_ids=None)),
             ("posv", dict(position=0, positions=_t(golden[f"rope_{style}_positions"].astype(np.int32), dev), position_ids=None)),
             ("pid", dict(position=0, positions=None, position_ids=_t(golden[f"rope_{style}_pid"].astype(np.int32), dev)))]
    for tag, kw in cases:
        qo = torch.empty(q.shape, dtype=torch.half, device=dev); ko = torch.empty(k.shape, dtype=torch.half, device=dev)
        ext.rope(_t(q, dev), qo, _t(k, dev), ko, _t(inv, dev), kw["position"], kw["positions"], kw["position_ids"], mode, 1.0)
        assert np.allclose(qo.float().cpu().numpy(), golden[f"rope_{style}_{tag}_q"].astype(np.float32), atol=3e-3, rtol=3e-3)
        assert np.allclose(ko.float().cpu().numpy(), golden[f"rope_{style}_{tag}_k"].astype(np.float32), atol=3e-3, rtol=3e-3)


@pytest.mark.parametrize("shape", [(1, 1, 32, 8, 128), (2, 7, 32, 8, 64), (1, 33, 4, 4, 256), (3, 5, 8, 2, 96)])
def test_rope_shapes_norm_inplace(dev, shape):
    # tests/test_rope.py:36-105 style sweep: (bsz, seq, heads_q, heads_k, head_dim), +- fused head norm, in place
    from exllamav3_amd import ext
    b, s, hq, hk, hd = shape
    rng = np.random.default_rng(hd + s)
    q = rng.standard_normal((b, s, hq, hd)).astype(np.float16)
    k = rng.standard_normal((b, s, hk, hd)).astype(np.float16)
    inv = (1.0 / (10000.0 ** (np.arange(0, hd, 2) / hd))).astype(np.float32)
    for mode in (1, 2):
        for with_norm in (False, True):
            qn = (1 + 0.1 * rng.standard_normal(hd)).astype(np.float16) if with_norm else None
            kn = (1 + 0.1 * rng.standard_normal(hd)).astype(np.float16) if with_norm else None
            rq, rk = o.rope(q, k, inv, position=11, rope_mode=mode, attn_factor=1.1, q_norm=qn, k_norm=kn, norm_eps=1e-6, norm_constant_bias=0.0)
            tq, tk = _t(q, dev), _t(k, dev)
            ext.rope(tq, tq, tk, tk, _t(inv, dev), 11, None, None, mode, 1.1, None if qn is None else _t(qn, dev),
                     None if kn is None else _t(kn, dev), 1e-6, 0.0)
            tol = 6e-3 if with_norm else 3e-3
            assert np.allclose(tq.float().cpu().numpy(), rq.astype(np.float32), atol=tol, rtol=tol)
            assert np.allclose(tk.float().cpu().numpy(), rk.astype(np.float32), atol=tol, rtol=tol)


@pytest.mark.parametrize("bits", range(2, 9))
def test_kv_quant_cont_bit_exact(dev, bits):
    from exllamav3_amd import ext
    rng = np.random.default_rng(bits)
    x = (rng.standard_normal((37, 1024)) * rng.uniform(0.1, 4.0, size=(37, 1))).astype(np.float16)
    x[3, :64] = 0                                              # an all-zero group pair
    pk_ref, sc_ref = o.kv_quant(x, bits)
    out = torch.zeros((37, 1024 // 32 * bits), dtype=torch.int32, device=dev)
    sc = torch.zeros((37, 32), dtype=torch.half, device=dev)
    ext.quant_cache_cont(_t(x, dev), out, sc)
    assert np.array_equal(sc.cpu().numpy().view(np.uint16), sc_ref.view(np.uint16))
    assert np.array_equal(out.cpu().numpy().view(np.uint32), pk_ref)
    y = torch.empty((37, 1024), dtype=torch.half, device=dev)
    ext.dequant_cache_cont(out, sc, y)
    y_ref = o.kv_dequant(pk_ref, sc_ref, bits)
    assert np.array_equal(y.cpu().numpy().view(np.uint16), y_ref.view(np.uint16))
    if bits == 8:       # tests/test_kv_quant.py tolerance (unit-normal data)
        xu = rng.standard_normal((8, 1024)).astype(np.float16)
        o8 = torch.zeros((8, 1024 // 32 * 8), dtype=torch.int32, device=dev); s8 = torch.zeros((8, 32), dtype=torch.half, device=dev)
        y8 = torch.empty((8, 1024), dtype=torch.half, device=dev)
        ext.quant_cache_cont(_t(xu, dev), o8, s8); ext.dequant_cache_cont(o8, s8, y8)
        assert np.allclose(y8.float().cpu().numpy(), xu.astype(np.float32), atol=0.08, rtol=0.01)


@pytest.mark.parametrize("bits", [2, 3, 4, 5, 8])
@pytest.mark.parametrize("a", [0.65, 0.3])
def test_kv_compander_against_oracle(dev, bits, a):
    """compand_a > 0 (cache/lmq.cuh through quant / dequant_cache_cont): the decoder is pure fp32 arithmetic -> dequantized values bit-exact
    vs the (reference-pinned) oracle; the encoder goes through cbrtf, whose device implementation may differ from libm's in the last ulp ->
    scales bit-exact, level indices identical except for at most a handful that land in the neighbouring cell."""
    from exllamav3_amd import ext
    rng = np.random.default_rng(40 + bits)
    x = (rng.standard_normal((41, 1024)) * rng.uniform(0.1, 4.0, size=(41, 1))).astype(np.float16)
    x[3, :64] = 0
    pk_ref, sc_ref = o.kv_quant(x, bits, a)
    out = torch.zeros((41, 1024 // 32 * bits), dtype=torch.int32, device=dev)
    sc = torch.zeros((41, 32), dtype=torch.half, device=dev)
    ext.quant_cache_cont(_t(x, dev), out, sc, a)
    assert np.array_equal(sc.cpu().numpy().view(np.uint16), sc_ref.view(np.uint16))
    # decoder: the oracle's words through the HIP dequantizer
    y = torch.empty((41, 1024), dtype=torch.half, device=dev)
    ext.dequant_cache_cont(_t(pk_ref.view(np.int32), dev), _t(sc_ref, dev), y, a)
    y_ref = o.kv_dequant(pk_ref, sc_ref, bits, a)
    assert np.array_equal(y.cpu().numpy().view(np.uint16), y_ref.view(np.uint16))
    # encoder: levels of the HIP words vs the oracle's, compared through the (exact) decoder in the rotated domain
    lv = lambda words: o.kv_dequant(words, np.ones_like(sc_ref), bits, a).astype(np.float32)     # unit scales: H32 of the levels
    mine = out.cpu().numpy().view(np.uint32)
    diff = (lv(mine) != lv(pk_ref)).reshape(-1, 32).any(-1)                # groups with any differing level
    assert diff.mean() < 2e-3, diff.mean()
    y2 = torch.empty_like(y)
    ext.dequant_cache_cont(out, sc, y2, a)                                  # and whatever differs is one cell off at most
    step = 2.0 / (1 << bits) * 1.5                                          # widest cell of the companded grid is < 1.5 midpoint cells
    assert np.abs(y2.float().cpu().numpy() - y_ref.astype(np.float32)).max() <= step * np.abs(x.astype(np.float32)).max() * 2


def test_kv_paged_sliding_window_window_scratch_and_cache_resident_input(dev):
    """The remaining arguments of the paged cache ops against the oracle (which is pinned against the reference kernels' own index arithmetic):
    sliding_window > 0 leaves exactly the rows the reference skips untouched; dequant_cache_paged_window writes the dense per-sequence scratch
    up to cache_seqlens + bonus_len; quant_cache_paged with in_contiguous=False reads the new tokens from the flat fp16 cache rows."""
    from exllamav3_amd import ext
    page, D, bsz, pps, bits = 256, 160, 2, 2, 4
    G = D // 32
    rng = np.random.default_rng(5)
    npages = bsz * pps + 1
    bt_np = rng.permutation(npages)[: bsz * pps].reshape(bsz, pps).astype(np.int32)
    seqlens = np.array([300, 131], np.int32)
    kq = rng.integers(0, 2 ** 32, size=(npages, page, G * bits), dtype=np.uint64).astype(np.uint32); vq = kq[::-1].copy()
    ks = rng.uniform(0.1, 4.0, size=(npages, page, G)).astype(np.float16); vs = ks[::-1].copy()
    T = lambda a: _t(a.view(np.int32) if a.dtype == np.uint32 else a, dev)
    for sw in (70, 33, 299):
        ko = torch.full((npages, page, D), float("nan"), dtype=torch.half, device=dev); vo = ko.clone()
        ext.dequant_cache_paged(T(kq), T(ks), ko, T(vq), T(vs), vo, T(seqlens), T(bt_np), page, sliding_window=sw)
        ko2 = np.full((npages, page, D), np.nan, np.float16); vo2 = ko2.copy()
        o.kv_dequant_paged(kq, ks, ko2, vq, vs, vo2, seqlens, bt_np, bits, bits, page, sliding_window=sw)
        assert np.array_equal(ko.cpu().numpy().view(np.uint16), ko2.view(np.uint16)) and np.array_equal(vo.cpu().numpy().view(np.uint16), vo2.view(np.uint16))
    ko = torch.full((bsz * pps, page, D), float("nan"), dtype=torch.half, device=dev); vo = ko.clone()
    ext.dequant_cache_paged_window(T(kq), T(ks), ko, T(vq), T(vs), vo, T(seqlens), T(bt_np), page, 7)
    ko2 = np.full((bsz * pps, page, D), np.nan, np.float16); vo2 = ko2.copy()
    o.kv_dequant_paged(kq, ks, ko2, vq, vs, vo2, seqlens, bt_np, bits, bits, page, bonus_len=7, compact_out=True)
    assert np.array_equal(ko.cpu().numpy().view(np.uint16), ko2.view(np.uint16)) and np.array_equal(vo.cpu().numpy().view(np.uint16), vo2.view(np.uint16))
    # in_contiguous = False: the new tokens already sit in the fp16 cache at their own rows
    kf = rng.standard_normal((npages, page, D)).astype(np.float16); vf = rng.standard_normal((npages, page, D)).astype(np.float16)
    n_new = 6
    mk = lambda: (np.zeros((npages, page, G * bits), np.uint32), np.zeros((npages, page, G), np.float16))
    (rk, rks), (rv, rvs) = mk(), mk()
    o.kv_quant_paged(kf, rk, rks, vf, rv, rvs, seqlens, bt_np, bits, bits, page, in_contiguous=False, seq_len=n_new)
    dk, dks, dv, dvs = (torch.zeros(a.shape, dtype=torch.int32 if a.dtype == np.uint32 else torch.half, device=dev) for a in (rk, rks, rv, rvs))
    ext.quant_cache_paged(T(kf), dk, dks, T(vf), dv, dvs, T(seqlens), T(bt_np), page, n_new, in_contiguous=False)
    assert np.array_equal(dk.cpu().numpy().view(np.uint32), rk) and np.array_equal(dv.cpu().numpy().view(np.uint32), rv)
    assert np.array_equal(dks.cpu().numpy().view(np.uint16), rks.view(np.uint16)) and np.array_equal(dvs.cpu().numpy().view(np.uint16), rvs.view(np.uint16))
    assert rk.any()


@pytest.mark.parametrize("kb,vb", [(8, 8), (4, 4), (6, 5), (2, 3)])
def test_kv_quant_paged_roundtrip(dev, kb, vb):
    """tests/test_kv_quant.py:21-143: paged quant -> dequant with a permuted page table and a 5-token append."""
    from exllamav3_amd import ext
    page, dim, bsz, pages_per_seq = 256, 1024, 2, 3
    total_pages = bsz * pages_per_seq + 2
    rng = np.random.default_rng(kb * 10 + vb)
    perm = rng.permutation(total_pages)[: bsz * pages_per_seq].astype(np.int32).reshape(bsz, pages_per_seq)
    bt = _t(perm, dev)
    G = dim // 32
    kq = torch.zeros((total_pages, page, G * kb), dtype=torch.int32, device=dev); ks = torch.zeros((total_pages, page, G), dtype=torch.half, device=dev)
    vq = torch.zeros((total_pages, page, G * vb), dtype=torch.int32, device=dev); vs = torch.zeros((total_pages, page, G), dtype=torch.half, device=dev)
    seqlens = np.array([0, 0], dtype=np.int32)
    lens = [300, 5]                                               # a 300-token prefill then a 5-token append
    full_k = [np.zeros((0, dim), np.float16) for _ in range(bsz)]; full_v = [np.zeros((0, dim), np.float16) for _ in range(bsz)]
    for n_new in lens:
        k_new = rng.standard_normal((bsz, n_new, dim)).astype(np.float16); v_new = rng.standard_normal((bsz, n_new, dim)).astype(np.float16)
        ext.quant_cache_paged(_t(k_new, dev), kq, ks, _t(v_new, dev), vq, vs, _t(seqlens, dev), bt, page, n_new)
        for b in range(bsz):
            full_k[b] = np.concatenate([full_k[b], k_new[b]]); full_v[b] = np.concatenate([full_v[b], v_new[b]])
        seqlens = seqlens + n_new
    ko = torch.zeros((total_pages, page, dim), dtype=torch.half, device=dev); vo = torch.zeros_like(ko)
    ext.dequant_cache_paged(kq, ks, ko, vq, vs, vo, _t(seqlens, dev), bt, page)
    ko, vo = ko.cpu().numpy(), vo.cpu().numpy()
    for b in range(bsz):
        for t in range(int(seqlens[b])):
            pg = perm[b, t // page]
            for full, got, bits in ((full_k[b], ko, kb), (full_v[b], vo, vb)):
                pk, sc = o.kv_quant(full[t:t + 1], bits)
                ref = o.kv_dequant(pk, sc, bits)[0]
                assert np.array_equal(got[pg, t % page].view(np.uint16), ref.view(np.uint16)), (b, t, bits)


@pytest.mark.parametrize("kb,vb", [(4, 4), (8, 3), (2, 6)])
@pytest.mark.parametrize("lens", [[1], [63, 64], [300, 1000, 77]])
@pytest.mark.parametrize("hd,hq,hkv", [(128, 8, 2), (64, 8, 2), (64, 32, 8)])
def test_attn_decode_qcache(dev, kb, vb, lens, hd, hq, hkv):
    """Decode attention straight from the quantized paged cache (rotated-domain scores / accumulation, flash-decoding splits) against the
    oracle attention over the dequantized cache; scattered pages, ragged lengths, GQA 4; head_dim 128 and 64 (two kv heads per 128-value
    block: Llama-3.2-1B's 32 q / 8 kv heads x 64)."""
    from exllamav3_amd import ext
    rng = np.random.default_rng(kb * 10 + vb + len(lens))
    bsz, page = len(lens), 256
    maxlen = max(lens)
    pps = (maxlen + page - 1) // page
    npages = bsz * pps + 3
    perm = rng.permutation(npages)[: bsz * pps].reshape(bsz, pps).astype(np.int32)          # scattered physical pages
    G = hkv * hd // 32
    k = (rng.standard_normal((bsz, pps * page, hkv * hd)) * 1.5).astype(np.float16)
    v = rng.standard_normal((bsz, pps * page, hkv * hd)).astype(np.float16)
    kq, ks = o.kv_quant(k, kb); vq, vs = o.kv_quant(v, vb)
    kc = np.zeros((npages, page, G * kb), dtype=np.uint32); ksc = np.zeros((npages, page, G), dtype=np.float16)
    vc = np.zeros((npages, page, G * vb), dtype=np.uint32); vsc = np.zeros((npages, page, G), dtype=np.float16)
    for b in range(bsz):
        for p in range(pps):
            kc[perm[b, p]] = kq[b, p * page:(p + 1) * page]; ksc[perm[b, p]] = ks[b, p * page:(p + 1) * page]
            vc[perm[b, p]] = vq[b, p * page:(p + 1) * page]; vsc[perm[b, p]] = vs[b, p * page:(p + 1) * page]
    q = rng.standard_normal((bsz, hq, hd)).astype(np.float16)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out = torch.full((bsz, hq, hd), float("nan"), dtype=torch.half, device=dev)
    ext.attn_decode_qcache(T(q), out, T(kc.view(np.int32)), T(ksc), T(vc.view(np.int32)), T(vsc), T(perm), T(np.array(lens, dtype=np.int32)), maxlen)
    kd = o.kv_dequant(kq, ks, kb).reshape(bsz, -1, hkv, hd); vd = o.kv_dequant(vq, vs, vb).reshape(bsz, -1, hkv, hd)
    ref = o.attn_decode_qcache(q, kd, vd, lens).astype(np.float32)
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() / np.sqrt((ref ** 2).mean()) < 1e-2


@pytest.mark.parametrize("kb,vb,hd,hq,hkv", [(4, 4, 128, 8, 2), (8, 3, 128, 4, 4), (4, 4, 64, 8, 2), (2, 6, 64, 32, 8)])
@pytest.mark.parametrize("lens,max_len", [([1], 1), ([63, 64], 64), ([300, 1000, 77], 1000), ([2500, 1, 700], 2560)])
def test_attn_decode_qcache_with_sinks(dev, kb, vb, hd, hq, hkv, lens, max_len):
    """Learned attention sinks (one fp32 logit per query head, gpt-oss style): the logit joins the softmax denominator at the final reduction and
    carries no value (modules/attention_fn/triton_paged.py:1030-1050; the torch restatement of the reference's own tests concatenates the sink
    logit, takes the softmax and drops its column: tests/test_dsa_kernels.py:54-62).  One-token contexts (a sink larger than the score halves the
    output), one split, many splits, the matrix-pipe kernel; head_dim 128 and 64; sinks from far below to far above the scores."""
    from exllamav3_amd import ext
    rng = np.random.default_rng(kb * 100 + vb * 10 + len(lens) + hd)
    bsz, page = len(lens), 256
    pps = (max_len + page - 1) // page
    npages = bsz * pps + 3
    perm = rng.permutation(npages)[: bsz * pps].reshape(bsz, pps).astype(np.int32)
    G = hkv * hd // 32
    k = (rng.standard_normal((bsz, pps * page, hkv * hd)) * 1.5).astype(np.float16)
    v = rng.standard_normal((bsz, pps * page, hkv * hd)).astype(np.float16)
    kq, ks = o.kv_quant(k, kb); vq, vs = o.kv_quant(v, vb)
    kc = np.zeros((npages, page, G * kb), dtype=np.uint32); ksc = np.zeros((npages, page, G), dtype=np.float16)
    vc = np.zeros((npages, page, G * vb), dtype=np.uint32); vsc = np.zeros((npages, page, G), dtype=np.float16)
    for b in range(bsz):
        for pg in range(pps):
            kc[perm[b, pg]] = kq[b, pg * page:(pg + 1) * page]; ksc[perm[b, pg]] = ks[b, pg * page:(pg + 1) * page]
            vc[perm[b, pg]] = vq[b, pg * page:(pg + 1) * page]; vsc[perm[b, pg]] = vs[b, pg * page:(pg + 1) * page]
    q = rng.standard_normal((bsz, hq, hd)).astype(np.float16)
    sinks = np.linspace(-6.0, 9.0, hq).astype(np.float32)                                      # scores here have a standard deviation of ~1.5
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    kd = o.kv_dequant(kq, ks, kb).reshape(bsz, -1, hkv, hd); vd = o.kv_dequant(vq, vs, vb).reshape(bsz, -1, hkv, hd)
    outs = {}
    for sk in (None, sinks):
        out = torch.full((bsz, hq, hd), float("nan"), dtype=torch.half, device=dev)
        ext.attn_decode_qcache(T(q), out, T(kc.view(np.int32)), T(ksc), T(vc.view(np.int32)), T(vsc), T(perm), T(np.array(lens, dtype=np.int32)), max_len,
                               sinks=None if sk is None else T(sk))
        ref = o.attn_decode_qcache(q, kd, vd, lens, sinks=sk).astype(np.float32)
        got = out.float().cpu().numpy()
        assert np.isfinite(got).all()
        assert np.abs(got - ref).max() / np.sqrt((o.attn_decode_qcache(q, kd, vd, lens).astype(np.float32) ** 2).mean()) < 1e-2
        outs[sk is None] = got
    # the sink only ever shrinks a head's output, by the factor sum / (sum + exp(sink - max))
    n0, n1 = np.abs(outs[True]).sum(-1), np.abs(outs[False]).sum(-1)
    assert (n1 <= n0 * 1.001 + 1e-3).all() and (n1[:, -1] < 0.9 * n0[:, -1]).all()


@pytest.mark.parametrize("waves", [0, 8])
@pytest.mark.parametrize("hd", [128, 64])
@pytest.mark.parametrize("hq,hkv", [(8, 2), (8, 1), (6, 2), (4, 4)])
@pytest.mark.parametrize("lens,max_len", [([2500, 1, 700], 2560), ([5000, 4096], 8192)])
def test_attn_decode_qcache_long_context_kernel(dev, hq, hkv, lens, max_len, hd, waves):
    """The matrix-pipe decode-attention kernel (head_dim 128, 4-bit K / V, length bound >= 2048): 16 tokens per wave step, scores and value
    products as matrix instructions on fp16 values dequantized in pair order, V gathered with the LDS transpose read; GQA 4 / 8 / 3 / 1, ragged
    lengths (one token, ends inside / on a 64-token step and a page), a length bound well above the lengths; against the oracle attention
    over the dequantized cache.  waves: 4 (0 = the default at these split lengths) or 8 waves per workgroup (round 5: two token-step chains
    per SIMD; a step then covers 128 tokens, waves past the end of a split idle)."""
    from exllamav3_amd import ext
    if hd == 64 and hkv % 2:
        pytest.skip("head_dim 64 needs whole 128-value kv blocks (an even number of kv heads)")
    # head_dim 64 (round 4): the same kernel on the two kv heads of a 128-value block, rows = the query heads of both (2 x 4 at most; (8, 2) and (4, 4)
    # take it, (6, 2) too: 3 heads per kv head)
    kb, vb, page = 4, 4, 256
    rng = np.random.default_rng(hq * 10 + hkv + len(lens))
    bsz = len(lens)
    pps = max_len // page
    used = (max(lens) + page - 1) // page
    npages = bsz * pps + 3
    perm = rng.permutation(npages)[: bsz * pps].reshape(bsz, pps).astype(np.int32)
    G = hkv * hd // 32
    k = (rng.standard_normal((bsz, used * page, hkv * hd)) * 1.5).astype(np.float16)
    v = rng.standard_normal((bsz, used * page, hkv * hd)).astype(np.float16)
    kq, ks = o.kv_quant(k, kb); vq, vs = o.kv_quant(v, vb)
    kc = np.zeros((npages, page, G * kb), dtype=np.uint32); ksc = np.zeros((npages, page, G), dtype=np.float16)
    vc = np.zeros((npages, page, G * vb), dtype=np.uint32); vsc = np.zeros((npages, page, G), dtype=np.float16)
    for b in range(bsz):
        for p in range(used):
            kc[perm[b, p]] = kq[b, p * page:(p + 1) * page]; ksc[perm[b, p]] = ks[b, p * page:(p + 1) * page]
            vc[perm[b, p]] = vq[b, p * page:(p + 1) * page]; vsc[perm[b, p]] = vs[b, p * page:(p + 1) * page]
    q = rng.standard_normal((bsz, hq, hd)).astype(np.float16)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out = torch.full((bsz, hq, hd), float("nan"), dtype=torch.half, device=dev)
    ext.set_attn_wide_waves(waves)
    try:
        ext.attn_decode_qcache(T(q), out, T(kc.view(np.int32)), T(ksc), T(vc.view(np.int32)), T(vsc), T(perm), T(np.array(lens, dtype=np.int32)), max_len)
    finally:
        ext.set_attn_wide_waves(0)
    kd = o.kv_dequant(kq, ks, kb).reshape(bsz, -1, hkv, hd); vd = o.kv_dequant(vq, vs, vb).reshape(bsz, -1, hkv, hd)
    ref = o.attn_decode_qcache(q, kd, vd, lens).astype(np.float32)
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() / np.sqrt((ref ** 2).mean()) < 1e-2


@pytest.mark.parametrize("hq,hkv", [(8, 2), (7, 1), (6, 2), (4, 4), (16, 2)])
@pytest.mark.parametrize("lens,max_len", [([1024, 1, 700, 257], 1024), ([130, 16], 130), ([1000], 1001), ([512, 513, 255, 256], 640)])
def test_attn_decode_qcache_short_contexts_ragged(dev, hq, hkv, lens, max_len):
    """Contexts up to 1024 tokens on the matrix-pipe decode-attention kernel (head_dim 128, 4-bit K / V; written for a one-launch variant -- one
    16-wave workgroup per (sequence, kv head), no merge launch -- that measured 2.44 vs 1.89 ms per step and was dropped: profiles/NOTES.md B 4.10): GQA 4 / 7 /
    3 / 1 / 8, ragged lengths: a full 1024, one token, ends on / next to a 256-token boundary and a 16-token wave tile, a bound above the lengths;
    scattered pages; against the oracle; the same bits on a second call."""
    from exllamav3_amd import ext
    hd, kb, vb, page = 128, 4, 4, 256
    rng = np.random.default_rng(hq * 10 + hkv + len(lens) + max_len)
    bsz = len(lens)
    pps = (max_len + page - 1) // page
    npages = bsz * pps + 3
    perm = rng.permutation(npages)[: bsz * pps].reshape(bsz, pps).astype(np.int32)
    G = hkv * hd // 32
    k = (rng.standard_normal((bsz, pps * page, hkv * hd)) * 1.5).astype(np.float16)
    v = rng.standard_normal((bsz, pps * page, hkv * hd)).astype(np.float16)
    kq, ks = o.kv_quant(k, kb); vq, vs = o.kv_quant(v, vb)
    kc = np.zeros((npages, page, G * kb), dtype=np.uint32); ksc = np.zeros((npages, page, G), dtype=np.float16)
    vc = np.zeros((npages, page, G * vb), dtype=np.uint32); vsc = np.zeros((npages, page, G), dtype=np.float16)
    for b in range(bsz):
        for p in range(pps):
            kc[perm[b, p]] = kq[b, p * page:(p + 1) * page]; ksc[perm[b, p]] = ks[b, p * page:(p + 1) * page]
            vc[perm[b, p]] = vq[b, p * page:(p + 1) * page]; vsc[perm[b, p]] = vs[b, p * page:(p + 1) * page]
    q = rng.standard_normal((bsz, hq, hd)).astype(np.float16)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out = torch.full((bsz, hq, hd), float("nan"), dtype=torch.half, device=dev)
    args = (T(q), out, T(kc.view(np.int32)), T(ksc), T(vc.view(np.int32)), T(vsc), T(perm), T(np.array(lens, dtype=np.int32)), max_len)
    ext.attn_decode_qcache(*args)
    kd = o.kv_dequant(kq, ks, kb).reshape(bsz, -1, hkv, hd); vd = o.kv_dequant(vq, vs, vb).reshape(bsz, -1, hkv, hd)
    ref = o.attn_decode_qcache(q, kd, vd, lens).astype(np.float32)
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() / np.sqrt((ref ** 2).mean()) < 1e-2
    out2 = torch.full_like(out, float("nan"))
    ext.attn_decode_qcache(args[0], out2, *args[2:])
    assert torch.equal(out, out2)


@pytest.mark.timeout(120)
def test_attn_decode_qcache_more_workgroups_than_the_wide_kernel_cap(dev):
    """bsz * kv blocks above the matrix-pipe kernel's workgroup cap (512) with head_dim 128, 4-bit K / V and a length bound >= 128: the split
    search of the wide route has to stop at one split (it used to spin forever on the host: ADVICE round 2) and the half-wave-per-token kernel
    takes the batch."""
    from exllamav3_amd import ext
    hd, hq, hkv, kb, vb, page = 128, 16, 8, 4, 4, 256
    rng = np.random.default_rng(77)
    bsz = 70                                                                                  # 70 * 8 = 560 workgroups per split
    lens = [int(x) for x in rng.integers(1, 200, bsz)]
    max_len = 200
    npages = bsz + 2
    perm = rng.permutation(npages)[:bsz].reshape(bsz, 1).astype(np.int32)
    G = hkv * hd // 32
    k = (rng.standard_normal((bsz, page, hkv * hd)) * 1.5).astype(np.float16)
    v = rng.standard_normal((bsz, page, hkv * hd)).astype(np.float16)
    kq, ks = o.kv_quant(k, kb); vq, vs = o.kv_quant(v, vb)
    kc = np.zeros((npages, page, G * kb), dtype=np.uint32); ksc = np.zeros((npages, page, G), dtype=np.float16)
    vc = np.zeros((npages, page, G * vb), dtype=np.uint32); vsc = np.zeros((npages, page, G), dtype=np.float16)
    for b in range(bsz):
        kc[perm[b, 0]] = kq[b]; ksc[perm[b, 0]] = ks[b]; vc[perm[b, 0]] = vq[b]; vsc[perm[b, 0]] = vs[b]
    q = rng.standard_normal((bsz, hq, hd)).astype(np.float16)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out = torch.full((bsz, hq, hd), float("nan"), dtype=torch.half, device=dev)
    ext.attn_decode_qcache(T(q), out, T(kc.view(np.int32)), T(ksc), T(vc.view(np.int32)), T(vsc), T(perm), T(np.array(lens, dtype=np.int32)), max_len)
    kd = o.kv_dequant(kq, ks, kb).reshape(bsz, -1, hkv, hd); vd = o.kv_dequant(vq, vs, vb).reshape(bsz, -1, hkv, hd)
    ref = o.attn_decode_qcache(q, kd, vd, lens).astype(np.float32)
    got = out.float().cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() / np.sqrt((ref ** 2).mean()) < 1e-2


@pytest.mark.parametrize("shape", [(1, 300, 32, 8, 128), (2, 17, 5, 3, 128), (1, 64, 40, 40, 128), (1, 16, 1, 0, 128)])
def test_rope_neox_head_dim_128_prefill_kernel(dev, shape):
    """>= 16 tokens, NEOX, head_dim 128, no head norm: the 16-byte-per-lane kernel (8 lanes per head); scalar position, per-sequence
    positions and position ids, in place and out of place, q only (heads_k = 0)."""
    from exllamav3_amd import ext
    b, s, hq, hk, hd = shape
    rng = np.random.default_rng(s + hq)
    q = rng.standard_normal((b, s, hq, hd)).astype(np.float16)
    k = rng.standard_normal((b, s, hk, hd)).astype(np.float16) if hk else None
    inv = (1.0 / (500000.0 ** (np.arange(0, hd, 2) / hd))).astype(np.float32)
    positions = rng.integers(0, 5000, size=b).astype(np.int32)
    pid = rng.integers(0, 100000, size=(b, s)).astype(np.int32)
    for kw in (dict(position=1234), dict(positions=positions), dict(position_ids=pid)):
        rq, rk = o.rope(q, k, inv, rope_mode=2, attn_factor=0.9, **kw)
        qo = torch.empty(q.shape, dtype=torch.half, device=dev)
        ko = torch.empty(k.shape, dtype=torch.half, device=dev) if hk else None
        ext.rope(_t(q, dev), qo, _t(k, dev) if hk else None, ko, _t(inv, dev), kw.get("position", 0),
                 _t(kw["positions"], dev) if "positions" in kw else None, _t(kw["position_ids"], dev) if "position_ids" in kw else None, 2, 0.9)
        assert np.allclose(qo.float().cpu().numpy(), rq.astype(np.float32), atol=3e-3, rtol=3e-3)
        if hk:
            assert np.allclose(ko.float().cpu().numpy(), rk.astype(np.float32), atol=3e-3, rtol=3e-3)
    tq = _t(q, dev)
    tk = _t(k, dev) if hk else None
    rq, rk = o.rope(q, k, inv, position=7, rope_mode=2)
    ext.rope(tq, tq, tk, tk, _t(inv, dev), 7, None, None, 2, 1.0)
    assert np.allclose(tq.float().cpu().numpy(), rq.astype(np.float32), atol=3e-3, rtol=3e-3)


def _rope_ex_case(dev, b, s, hq, hk, hd, *, mode=2, partial=None, rotate_dims=1, rotate_offset=0, pos_mode="position", table=0, norm=None, bias=0.0,
                  post_norm=False, l4=None, attn_factor=1.0, pad=0, in_place=False, seed=0):
    """One call of ext.rope through exl3_rope_ex against the oracle (which the reference's own tests/test_rope.py judges on the CPU:
    tests/test_oracle_pins.py::test_rope_full_argument_list_against_the_references_own_tests)."""
    from exllamav3_amd import ext
    rng = np.random.default_rng(seed + hd + 7 * s)
    partial = partial or hd // rotate_dims
    # heads as the trailing hd columns of (hd + pad)-wide heads: q.stride(2) = hd + pad (rope.cu:351-353)
    qw = rng.standard_normal((b, s, hq, hd + pad)).astype(np.float16)
    kw_ = rng.standard_normal((b, s, hk, hd + pad)).astype(np.float16) if hk else None
    q, k = qw[..., pad:], (kw_[..., pad:] if hk else None)
    inv = (1.0 / (10000.0 ** (np.arange(0, partial, 2) / partial))).astype(np.float32)
    kwargs = dict(position=0, positions=None, position_ids=None)
    if pos_mode == "position":
        kwargs["position"] = 19
    elif pos_mode == "positions":
        kwargs["positions"] = rng.integers(0, 49, size=b).astype(np.int32)
    elif pos_mode == "ids":
        kwargs["position_ids"] = rng.integers(0, 117, size=(b, s)).astype(np.int32)
    else:                                                   # one position per rotated sub-range
        kwargs["position_ids"] = rng.integers(0, 117, size=(b, s, rotate_dims)).astype(np.int32)
    inv_arg = inv
    if table:
        # angle table [pos][pairs] (2-D) or [batch][pos][pairs] (3-D), e.g. per-sequence frequency scaling
        base = np.arange(160, dtype=np.float32)[:, None] * inv[None, :]
        inv_arg = base if table == 2 else np.stack([base * np.float32(1.0 + 0.25 * i) for i in range(b)])
    qn = kn = None
    if norm:
        qn = (1 + 0.1 * rng.standard_normal(hd)).astype(np.float32)
        kn = (1 + 0.1 * rng.standard_normal(hd)).astype(np.float32)
        if norm == "fp16":
            qn, kn = qn.astype(np.float16), kn.astype(np.float16)
    l4_beta, l4_orig = l4 if l4 else (0.0, 1)
    rq, rk = o.rope(q, k, inv_arg, rope_mode=mode, attn_factor=attn_factor, q_norm=qn, k_norm=kn, norm_eps=1e-6, norm_constant_bias=bias,
                    l4_beta=l4_beta, l4_orig=l4_orig, post_rope_norm=post_norm, rotate_dims=rotate_dims, rotate_offset=rotate_offset,
                    norm_bf16=(norm == "bf16"), **kwargs)
    tqw, tkw = _t(qw, dev), (_t(kw_, dev) if hk else None)
    tq, tk = tqw[..., pad:], (tkw[..., pad:] if hk else None)
    oq = tq if in_place else torch.zeros_like(tqw)[..., pad:]
    ok_ = (tk if in_place else torch.zeros_like(tkw)[..., pad:]) if hk else None
    nt = lambda w: None if w is None else (_t(w, dev).to(torch.bfloat16) if norm == "bf16" else _t(w, dev))
    ext.rope(tq, oq, tk, ok_, _t(inv_arg, dev), kwargs["position"], None if kwargs["positions"] is None else _t(kwargs["positions"], dev),
             None if kwargs["position_ids"] is None else _t(kwargs["position_ids"], dev), mode, attn_factor, nt(qn), nt(kn), 1e-6, bias,
             l4_beta, l4_orig, post_norm, rotate_dims, rotate_offset)
    tol = 6e-3 if (norm or post_norm) else 3e-3
    assert np.allclose(oq.float().cpu().numpy(), rq.astype(np.float32), atol=tol, rtol=tol)
    if hk:
        assert np.allclose(ok_.float().cpu().numpy(), rk.astype(np.float32), atol=tol, rtol=tol)
    if not in_place:                                        # inputs untouched, the padding columns of the outputs never written
        assert torch.equal(tqw.cpu(), torch.from_numpy(qw))
    return oq, ok_


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("hd,partial", [(128, 64), (128, 32), (96, 48), (256, 64), (80, 40), (64, 64)])
def test_rope_partial_rotary_and_nanochat(dev, mode, hd, partial):
    """Rotated width smaller than the head (rope.cu:154-160: the tail passes through), the three pair conventions, every position mode."""
    for pos_mode in ("position", "positions", "ids"):
        _rope_ex_case(dev, 2, 9, 8, 2, hd, mode=mode, partial=partial, pos_mode=pos_mode, attn_factor=1.07, seed=mode)
    _rope_ex_case(dev, 1, 1, 28, 7, hd, mode=mode, partial=partial, norm="fp16", in_place=True)
    _rope_ex_case(dev, 1, 3, 4, 0, hd, mode=mode, partial=partial)                      # q only


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("norm", [None, "fp16", "bf16"])
@pytest.mark.parametrize("in_place", [False, True])
def test_rope_multidim_position_ids(dev, mode, norm, in_place):
    """tests/test_rope.py:108-230 (the reference's two-axis rope: head_dim 72 = 2 x 36, 3-D position ids), plus 3 and 4 sub-ranges and a
    2-D position tensor shared by the sub-ranges."""
    _rope_ex_case(dev, 2, 280, 16, 16, 72, mode=mode, rotate_dims=2, pos_mode="ids3", norm=norm, in_place=in_place)
    _rope_ex_case(dev, 1, 5, 8, 2, 96, mode=mode, rotate_dims=3, pos_mode="ids3", norm=norm, in_place=in_place)
    _rope_ex_case(dev, 3, 2, 4, 4, 128, mode=mode, rotate_dims=4, pos_mode="ids3", norm=norm, in_place=in_place)
    _rope_ex_case(dev, 2, 6, 4, 2, 64, mode=mode, rotate_dims=2, pos_mode="ids", norm=norm, in_place=in_place)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("partial", [128, 64])
def test_rope_llama4_query_scale(dev, mode, partial):
    """tests/test_rope.py:233-290: keys bitwise unscaled, the whole query head (also its non-rotated part) scaled by 1 + beta ln(1 + pos // orig);
    the window straddles the first scale step."""
    from exllamav3_amd import ext
    beta, orig = 0.1, 16384
    b, s, hq, hk, hd = 2, 16, 8, 2, 128
    rng = np.random.default_rng(partial)
    q = rng.standard_normal((b, s, hq, hd)).astype(np.float16)
    k = rng.standard_normal((b, s, hk, hd)).astype(np.float16)
    inv = (1.0 / (1000000.0 ** (np.arange(0, partial, 2) / partial))).astype(np.float32)
    pos = orig - s // 2
    outs = {}
    for bt in (beta, 0.0):
        oq, ok_ = torch.zeros((b, s, hq, hd), dtype=torch.half, device=dev), torch.zeros((b, s, hk, hd), dtype=torch.half, device=dev)
        ext.rope(_t(q, dev), oq, _t(k, dev), ok_, _t(inv, dev), pos, None, None, mode, 1.0, None, None, 1e-6, 0.0, bt, orig, False, 1, 0)
        outs[bt] = (oq.cpu(), ok_.cpu())
        rq, rk = o.rope(q, k, inv, position=pos, rope_mode=mode, l4_beta=bt, l4_orig=orig)
        assert np.allclose(oq.float().cpu().numpy(), rq.astype(np.float32), atol=3e-3, rtol=3e-3)
    assert torch.equal(outs[beta][1], outs[0.0][1])
    scale = 1.0 + beta * np.log(1.0 + ((pos + np.arange(s)) // orig).astype(np.float32))
    assert scale.min() == 1.0 and scale.max() > 1.0
    expect = (outs[0.0][0].float().numpy() * scale.reshape(1, s, 1, 1)).astype(np.float16).astype(np.float32)
    assert np.allclose(outs[beta][0].float().numpy(), expect, atol=3e-3, rtol=3e-3)
    _rope_ex_case(dev, 2, 5, 8, 2, 128, mode=mode, partial=partial, pos_mode="positions", l4=(0.2, 8), norm="fp16", post_norm=True)


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_rope_post_norm_tables_offsets_strides(dev, mode):
    """The remaining arguments: unweighted norm after the rotation, angle tables (2-D shared / 3-D per sequence), a rotate offset (rope.cu:363),
    heads that are the trailing columns of wider heads (rope.cu:351-353), bf16 norm weights with a constant bias, head_dim 512."""
    _rope_ex_case(dev, 2, 7, 8, 2, 128, mode=mode, post_norm=True)
    _rope_ex_case(dev, 2, 7, 8, 2, 128, mode=mode, post_norm=True, norm="bf16", bias=1.0)
    _rope_ex_case(dev, 1, 11, 4, 4, 64, mode=mode, table=2, pos_mode="ids")
    _rope_ex_case(dev, 3, 4, 4, 2, 128, mode=mode, table=3, pos_mode="positions", partial=64)
    _rope_ex_case(dev, 2, 3, 8, 8, 128, mode=mode, partial=32, rotate_offset=96)                         # rotate the LAST 32 columns
    _rope_ex_case(dev, 2, 3, 8, 8, 192, mode=mode, partial=64, rotate_offset=64, norm="fp16")
    _rope_ex_case(dev, 2, 5, 6, 3, 64, mode=mode, pad=448)                                               # rope on head[..., -64:] of 512-wide heads
    _rope_ex_case(dev, 1, 4, 2, 1, 64, mode=mode, pad=64, in_place=True, partial=32)
    _rope_ex_case(dev, 1, 2, 3, 1, 512, mode=mode, norm="fp16")
    _rope_ex_case(dev, 1, 2, 3, 1, 512, mode=mode, partial=128, rotate_dims=4, pos_mode="ids3")


def test_rope_argument_checks(dev):
    """The host checks of rope.cu:361-420 raise (RuntimeError) instead of launching."""
    from exllamav3_amd import ext
    q = torch.zeros((1, 2, 4, 128), dtype=torch.half, device=dev)
    inv32 = torch.ones(32, dtype=torch.float, device=dev)
    ids = torch.zeros((1, 2), dtype=torch.int32, device=dev)
    with pytest.raises(RuntimeError, match="rotate_dims"):
        ext.rope(q, q, None, None, inv32, 0, None, None, 2, 1.0, None, None, 1e-6, 0.0, 0.0, 1, False, 3, 0)           # 128 != 64 * 3
    with pytest.raises(RuntimeError, match="rotate_offset"):
        ext.rope(q, q, None, None, inv32, 0, None, None, 2, 1.0, None, None, 1e-6, 0.0, 0.0, 1, False, 1, 96)
    with pytest.raises(RuntimeError, match="invalid arguments"):
        ext.rope(q, q, None, None, inv32, 0, torch.zeros(1, dtype=torch.int32, device=dev), ids, 2, 1.0)
    with pytest.raises(RuntimeError, match="position_ids"):
        ext.rope(q, q, None, None, inv32, 0, None, torch.zeros((1, 2, 3), dtype=torch.int32, device=dev), 2, 1.0)
    with pytest.raises(RuntimeError, match="rope_mode"):
        ext.rope(q, q, None, None, inv32, 0, None, None, 4, 1.0)
