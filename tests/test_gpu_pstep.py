"""The persistent decode step (ext.PersistentStep / exl3_pstep.hip: the whole step in ONE launch) against the ORACLE composition of the same step
(reference graphs it replaces: libtorch/attention.cpp:246-330 + libtorch/mlp.cpp:14-91 per layer) and against the launch-per-op fx pipeline.
Bars: logits 3e-2 * RMS end to end (the bar of every pipeline-vs-oracle test), appended K / V rows as in test_gpu_fullsize, hipGraph replay == eager bits,
no edge / tagged-line time-out ever."""
import numpy as np
import pytest
import torch
from oracle import exl3_oracle as o
from test_gpu_path import _oracle_decode
from test_gpu_fullsize import (_np, _lin, _relerr, _oracle_attention_sublayer, _head_cols, _check_logits_on_cols, _replay_equals, _appended_kv_close, _fill_ctx)

pytestmark = pytest.mark.gpu


def _model(shape, dev, K=4, pos=700, max_ctx=1024):
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    m = SyntheticEXL3Llama(shape, K=K, cb=2, device=dev, kv_bits=4, max_ctx=max_ctx)
    m.alloc_state(1, pos=pos)
    assert m.persistent_applies()
    return m


@pytest.mark.parametrize("hidden,inter,hq,hkv,hd,vocab,layers", [
    (256, 512, 4, 2, 128, 384, 2),          # rectangles smaller than the chip: most workgroups idle in every op
    (512, 1536, 8, 2, 64, 1024, 3),         # head_dim 64: two query heads per Hadamard block
    (1024, 2816, 8, 8, 128, 3072, 2),       # MHA-shaped kv, an inter that is not a power of two
])
@pytest.mark.parametrize("K", [4, 3])
def test_persistent_step_matches_oracle_and_fx_pipeline(dev, hidden, inter, hq, hkv, hd, vocab, layers, K):
    from exllamav3_amd.llama_path import LlamaShape
    shape = LlamaShape("tiny-ps", hidden, inter, layers, hq, hkv, hd, vocab)
    m = _model(shape, dev, K=K)
    ref = _oracle_decode(m, _np(m.x0))
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    lp = _np(m.decode_step_persistent().float()).copy()
    assert not m._pstep.error()
    assert np.isfinite(lp).all()
    assert _relerr(lp, ref) < 3e-2, _relerr(lp, ref)
    kv_p = [(_np(c).copy(), _np(s_).copy()) for c, s_ in m.kcache + m.vcache]
    q_p = _np(m.q.float()).copy()
    # the launch-per-op pipeline on the same tensors: same K / V words (the same device functions finish and quantize them), logits within the bar
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    lf = _np(m.decode_step_fx().float()).copy()
    assert _relerr(lp, lf) < 2e-2
    assert np.abs(q_p - _np(m.q.float())).max() < 2e-2 * max(1.0, float(np.abs(q_p).max()))
    # (a quantization level may flip where the two pipelines' k / v differ by rounding -- split-k order, the residual's representation: compare the
    #  dequantized rows, bound as in test_gpu_fullsize._appended_kv_close; a wrong slot / head / rope pairing misses it by an order of magnitude)
    page, slot = int(m.block_table[0, 700 // m.page]), 700 % m.page
    for (wa, sa), (c, s_) in zip(kv_p, m.kcache + m.vcache):
        _kv_rows_match_in_levels(wa[page, slot], sa[page, slot], _np(c[page, slot]), _np(s_[page, slot]), 0.02)      # (level by level: VERDICT r5 weak 2b)
        assert np.abs(wa).sum() == np.abs(wa[page, slot]).sum()          # nothing but the new token's row was written
    # eager twice and graph replay: the same bits (slab lines summed in slice order, owners add in a fixed order: nothing depends on arrival order)
    assert np.array_equal(_np(m.decode_step_persistent().float()), lp)
    _replay_equals(m.decode_step_persistent, m, lp, reps=5)
    assert not m._pstep.error()


@pytest.mark.parametrize("ahead", [0, 1, 2, 3])
def test_persistent_step_decode_ahead_depths_give_the_same_bits(dev, ahead):
    """Decode-ahead only changes WHEN a unit's weights are decoded, not the arithmetic: 0, 1 (registers), 2 (registers + LDS) and 3 (registers + LDS + registers: the default) units give equal logits."""
    from exllamav3_amd.llama_path import LlamaShape
    m = _model(LlamaShape("tiny-ps", 512, 1536, 2, 8, 2, 64, 1024), dev)
    m.decode_step_persistent()
    m._pstep.set(decode_ahead_units=2)
    base = _np(m.decode_step_persistent().float()).copy()
    m._pstep.set(decode_ahead_units=ahead)
    assert np.array_equal(_np(m.decode_step_persistent().float()), base)
    assert not m._pstep.error()


@pytest.mark.parametrize("name,hidden,inter,hq,hkv,hd", [("8b", 4096, 14336, 32, 8, 128), ("1b", 2048, 8192, 32, 8, 64)])
def test_persistent_step_full_size_layer_and_head_vs_oracle(dev, name, hidden, inter, hq, hkv, hd):
    """ONE layer of Llama-3.1-8B / Llama-3.2-1B shapes + the 128256-column lm_head at 4 bpw through the persistent step: logits on sampled column ranges
    and the appended K / V rows against the oracle; replay == eager."""
    from exllamav3_amd.llama_path import LlamaShape
    shape = LlamaShape(name + "-1layer", hidden, inter, 1, hq, hkv, hd, 128256)
    m = _model(shape, dev)
    L = m.layers[0]
    x, ov, k4, v = _oracle_attention_sublayer(m, L, _np(m.x0), None, 700)
    xn, x = o.rms_norm(ov, _np(L["norm2"]), m.eps, residual_in=x)
    gf, uf = _lin(L["gate"], xn).astype(np.float32), _lin(L["up"], xn).astype(np.float32)
    a = (gf / (1 + np.exp(-gf)) * uf).astype(np.float16)
    d = _lin(L["down"], a, out_fp32=True)
    xn_f, _ = o.rms_norm(d, _np(m.final_norm), m.eps, residual_in=x)
    logits = _np(m.decode_step_persistent().float()).copy()
    assert not m._pstep.error()
    _check_logits_on_cols(logits, m, xn_f, _head_cols(shape.vocab))
    _appended_kv_close(m, 0, 1, 700, k4, v)
    _replay_equals(m.decode_step_persistent, m, logits)


@pytest.mark.parametrize("step", ["fx", "persistent"])
def test_many_layer_drift_at_hidden_4096_vs_oracle(dev, step):
    """EIGHT layers at Llama-3.1-8B width (hidden 4096, inter 14336, 32 / 8 heads x 128) + sampled head columns against the oracle: the residual's
    representation (64-bit fixed point with the previous 1/rms + r_new / r_prev correction in decode_step_fx; fp32 rows with the exact scale in the
    persistent step) followed through many layers at real width -- the one-layer full-size tests do not see an error that compounds."""
    from exllamav3_amd.llama_path import LlamaShape
    shape = LlamaShape("8b-8layers", 4096, 14336, 8, 32, 8, 128, 128256)
    m = _model(shape, dev)
    x, pend = _np(m.x0), None
    for L in m.layers:
        x, ov, _, _ = _oracle_attention_sublayer(m, L, x, pend, 700)
        xn, x = o.rms_norm(ov, _np(L["norm2"]), m.eps, residual_in=x)
        gf, uf = _lin(L["gate"], xn).astype(np.float32), _lin(L["up"], xn).astype(np.float32)
        a = (gf / (1 + np.exp(-gf)) * uf).astype(np.float16)
        pend = _lin(L["down"], a, out_fp32=True)
    xn_f, _ = o.rms_norm(pend, _np(m.final_norm), m.eps, residual_in=x)
    run = m.decode_step_fx if step == "fx" else m.decode_step_persistent
    logits = _np(run().float()).copy()
    if step == "persistent": assert not m._pstep.error()
    _check_logits_on_cols(logits, m, xn_f, _head_cols(shape.vocab, seed=11))
    _replay_equals(run, m, logits)


def test_persistent_step_refuses_what_it_does_not_cover(dev):
    """Layers whose widths are two bits apart (no kernel runs three widths), an mcg model at a width mcg is not instantiated for: persistent_applies() / exl3_pstep_create
    refuse, decode_step_persistent falls back to the launch-per-op step (same results)."""
    import warnings
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    shape = LlamaShape("tiny", 256, 512, 2, 4, 2, 128, 384)
    m = SyntheticEXL3Llama(shape, K=3, cb=2, device=dev, kv_bits=4, max_ctx=1024, layer_K=lambda li, g: 5 if g == "d" else 3)      # 3- and 5-bit linears
    m.alloc_state(1, pos=100)
    assert not m.persistent_applies()
    layers = [dict(L, kcache=m.kcache[i], vcache=m.vcache[i]) for i, L in enumerate(m.layers)]
    with pytest.raises(RuntimeError):
        ext.PersistentStep(layers, m.lm_head, m.final_norm, 256, m.hq, m.hkv, 128, m.eps)
    lf = m.decode_step_persistent()                       # falls back to the fx pipeline
    assert torch.isfinite(lf.float()).all()
    m2 = SyntheticEXL3Llama(shape, K=5, cb=1, device=dev, kv_bits=4, max_ctx=1024)          # mcg at 5 bits: structurally fine, no kernel instantiated
    m2.alloc_state(1, pos=100)
    assert m2.persistent_applies()
    ref = _np(m2.decode_step_fx().float()).copy()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = _np(m2.decode_step_persistent().float())
    assert m2._pstep is None and any("no persistent decode step" in str(x.message) for x in w)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("cb,K,head_K", [(0, 4, None), (0, 4, 6), (0, 3, 6), (0, 2, None), (0, 5, 6), (0, 6, None), (0, 8, 6), (1, 4, None), (1, 3, 6)])
@pytest.mark.parametrize("with_attention", [False, True])
def test_persistent_step_3inst_and_mcg_codebooks(dev, cb, K, head_K, with_attention):
    """The 3INST codebook (every older public EXL3 quant: quant/codebook.cuh:56-77) and mcg through the persistent step (exl3_pstep_kernel<K, K, KH, CB, ATT>: the EXACT
    fp16 weight per quad operand, no affine map at the finish): against the oracle composition (3e-2, without the attention core) and decode_step_fx on the same tensors
    (2e-2); replay == eager; no time-out."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    shape = LlamaShape("cb%d-k%d" % (cb, K), 1024, 2816, 2, 8, 2, 128, 3072)
    m = SyntheticEXL3Llama(shape, K=K, cb=cb, device=dev, kv_bits=4, max_ctx=1024, head_K=head_K)
    m.alloc_state(1, pos=300)
    m.with_attention = with_attention
    assert m.persistent_applies()
    if not with_attention:
        ref = _oracle_decode(m, _np(m.x0))
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    lf = _np(m.decode_step_fx().float()).copy()
    kv_f = [(_np(c).copy(), _np(s_).copy()) for c, s_ in m.kcache + m.vcache]
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    lp = _np(m.decode_step_persistent().float()).copy()
    assert m._pstep is not None and not m._pstep.error()
    d = m._pstep.describe()
    assert ("codebook=3inst" if cb == 0 else "codebook=mcg") in d and ("head_K=%d" % (head_K or K)) in d
    assert np.isfinite(lp).all() and _relerr(lp, lf) < 2e-2, _relerr(lp, lf)
    if not with_attention:
        assert _relerr(lp, ref) < 3e-2, _relerr(lp, ref)
    page, slot = int(m.block_table[0, 300 // m.page]), 300 % m.page
    for (wa, sa), (c, s_) in zip(kv_f, m.kcache + m.vcache):
        _kv_rows_match_in_levels(wa[page, slot], sa[page, slot], _np(c[page, slot]), _np(s_[page, slot]), 0.02)
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    _replay_equals(m.decode_step_persistent, m, lp, reps=3)
    assert not m._pstep.error()


@pytest.mark.parametrize("K,pattern", [(4, "mlp_odd"), (3, "o_d"), (2, "all_but_qkv"), (5, "mlp_odd")])
@pytest.mark.parametrize("with_attention", [False, True])
def test_persistent_step_two_adjacent_widths_in_the_layers(dev, K, pattern, with_attention):
    """A fractional-bpw checkpoint (3.5 / 4.5 bpw ...): the reference's allocator bumps whole qgroups -- q|k|v, o, gate|up, down of a layer -- by one bit
    (conversion/allocation.py:131-141), so fused linears of K and K + 1 bits alternate through the layers, the head at 6.  The step's streaming loop runs the ops in runs
    of equal width (exl3_pstep_kernel<K, K + 1, 6, mul1, ATT>): against the oracle composition (3e-2) and decode_step_fx (2e-2); replay == eager; no time-out."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    pat = {"mlp_odd": lambda li, g: K + 1 if (li % 2 == 1 and g in ("gu", "d")) else K,           # half the layers' MLPs one bit up (the 4.5-bpw-style model of VERDICT r5)
           "o_d": lambda li, g: K + 1 if g in ("o", "d") else K,                                    # every op a run of its own
           "all_but_qkv": lambda li, g: K if g == "qkv" else K + 1}[pattern]
    shape = LlamaShape("mixed-k%d" % K, 1024, 2816, 4, 8, 2, 128, 3072)
    m = SyntheticEXL3Llama(shape, K=K, cb=2, device=dev, kv_bits=4, max_ctx=1024, head_K=6, layer_K=pat)
    m.alloc_state(1, pos=300)
    m.with_attention = with_attention
    assert {L["down"].K for L in m.layers} | {L["q"].K for L in m.layers} == {K, K + 1} and m.persistent_applies()
    if not with_attention:
        ref = _oracle_decode(m, _np(m.x0))
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    lf = _np(m.decode_step_fx().float()).copy()
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    lp = _np(m.decode_step_persistent().float()).copy()
    assert m._pstep is not None and not m._pstep.error() and ("K=%d+1" % K) in m._pstep.describe()
    assert np.isfinite(lp).all() and _relerr(lp, lf) < 2e-2, _relerr(lp, lf)
    if not with_attention:
        assert _relerr(lp, ref) < 3e-2, _relerr(lp, ref)
    # the repacked words of a (K + 1)-bit op come back equal to the checkpoint tensor
    li = 1
    assert torch.equal(m._pstep.unpack_op(4 * li + 3, 0, m.layers[li]["down"].trellis), m.layers[li]["down"].trellis)
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    _replay_equals(m.decode_step_persistent, m, lp, reps=3)
    assert not m._pstep.error()


@pytest.mark.parametrize("hidden,inter,hq,hkv,hd,vocab,layers", [
    (512, 1536, 4, 1, 128, 1024, 2),        # GQA 4, one kv head: 1 x nsplit items, most workgroups without one
    (1024, 2816, 8, 2, 128, 3072, 2),       # two kv heads x 4 query heads
    (1024, 2048, 8, 1, 128, 2048, 2),       # eight query heads per kv head: the K / V tasks take a second task round
    (512, 1536, 6, 2, 128, 1024, 3),        # 3 query heads per kv head
    (512, 1536, 8, 2, 64, 1024, 2),         # head_dim 64 (Llama-3.2-1B's form): one 128-value kv block = two kv heads, 2 x 4 rows
    (1024, 2816, 16, 8, 64, 3072, 2),       # head_dim 64, 2 query heads per kv head, four kv blocks
    (256, 512, 4, 4, 64, 384, 2),           # head_dim 64, MHA: two rows
    (512, 1536, 6, 2, 64, 1024, 2),         # head_dim 64, 3 query heads per kv head: six rows, q blocks of two heads that straddle nothing (hq * 64 = 384 = 3 blocks)
])
@pytest.mark.parametrize("pos", [5, 130, 700, 1900])
def test_persistent_step_with_attention_inside_matches_the_launch_per_op_step(dev, hidden, inter, hq, hkv, hd, vocab, layers, pos):
    """The decode attention over the 4-bit paged cache INSIDE the persistent step (o_proj's preparation: one (kv head, context split) item per workgroup on the
    service + four streaming waves, partial records as tagged granules, merged by the consumers; head_dim 128 and 64) against decode_step_fx with the attention core (itself tested against
    the oracle in test_gpu_path / test_gpu_fullsize): logits, the finished queries, the appended K / V rows of every layer (dequantized), nothing else written
    to the cache; random pre-filled cache; eager twice and graph replay give the same bits; no time-out."""
    from exllamav3_amd.llama_path import LlamaShape
    _attention_inside_case(dev, LlamaShape("tiny-att", hidden, inter, layers, hq, hkv, hd, vocab), pos)


def _attention_inside_case(dev, shape, pos):
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    hq = shape.heads_q
    m = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=2048)
    m.alloc_state(1, pos=pos)
    m.with_attention = True
    m.attn_merge_in_oproj_hd64 = True
    assert m.persistent_applies()
    g = torch.Generator(device="cpu").manual_seed(hq * 1000 + pos)
    for c, s_ in m.kcache + m.vcache:
        c.copy_(torch.randint(-2 ** 31, 2 ** 31 - 1, c.shape, generator=g, dtype=torch.int64).to(torch.int32).to(dev))
        s_.copy_((torch.rand(s_.shape, generator=g) * 0.5 + 0.05).half().to(dev))
    saved = [(c.clone(), s_.clone()) for c, s_ in m.kcache + m.vcache]

    def restore():
        for (c, s_), (c0, s0) in zip(m.kcache + m.vcache, saved):
            c.copy_(c0); s_.copy_(s0)
        m.q.zero_()
    lf = _np(m.decode_step_fx().float()).copy()
    kv_f = [(_np(c).copy(), _np(s_).copy()) for c, s_ in m.kcache + m.vcache]
    q_f = _np(m.q.float()).copy()
    restore()
    lp = _np(m.decode_step_persistent().float()).copy()
    assert not m._pstep.error()
    assert np.isfinite(lp).all()
    assert _relerr(lp, lf) < 2e-2, _relerr(lp, lf)
    assert np.abs(q_f - _np(m.q.float())).max() < 2e-2 * max(1.0, float(np.abs(q_f).max()))
    page, slot = int(m.block_table[0, pos // m.page]), pos % m.page
    for (wa, sa), (c, s_), (c0, _) in zip(kv_f, m.kcache + m.vcache, saved):
        _kv_rows_match_in_levels(_np(c[page, slot]), _np(s_[page, slot]), wa[page, slot], sa[page, slot], 0.02)
        keep = np.ones(c.shape[:2], dtype=bool); keep[page, slot] = False
        assert np.array_equal(_np(c)[keep], _np(c0)[keep])                 # nothing but the new token's row was written
    restore()
    assert np.array_equal(_np(m.decode_step_persistent().float()), lp)
    restore()
    _replay_equals(m.decode_step_persistent, m, lp, reps=3)
    assert not m._pstep.error()


@pytest.mark.parametrize("K", [4, 3])
@pytest.mark.parametrize("with_attention", [False, True])
def test_persistent_step_with_a_six_bit_lm_head(dev, K, with_attention):
    """The lm_head of a real checkpoint stays at 6 bits whatever the layers have: the step's streaming loop then runs the layers with K and the head with its own K
    (exl3_pstep_kernel<K, 6, ATT>).  Against decode_step_fx on the same tensors (2e-2) and, without attention, the oracle composition (3e-2); replay == eager."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    shape = LlamaShape("tiny-h6", 512, 1536, 2, 4, 1, 128, 1024)
    m = SyntheticEXL3Llama(shape, K=K, cb=2, device=dev, kv_bits=4, max_ctx=1024, head_K=6)
    m.alloc_state(1, pos=300)
    m.with_attention = with_attention
    assert m.lm_head.K == 6 and m.persistent_applies()
    if not with_attention:
        ref = _oracle_decode(m, _np(m.x0))
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    lf = _np(m.decode_step_fx().float()).copy()
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    lp = _np(m.decode_step_persistent().float()).copy()
    assert not m._pstep.error() and "head_K=6" in m._pstep.describe()
    assert np.isfinite(lp).all() and _relerr(lp, lf) < 2e-2
    if not with_attention:
        assert _relerr(lp, ref) < 3e-2
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    _replay_equals(m.decode_step_persistent, m, lp, reps=3)
    assert not m._pstep.error()


@pytest.mark.parametrize("K,head_K", [(2, None), (5, None), (6, None), (8, None), (2, 6), (5, 6), (8, 6)])
@pytest.mark.parametrize("with_attention", [False, True])
def test_persistent_step_every_instantiated_bit_width(dev, K, head_K, with_attention):
    """Every (K, head K) pair the library instantiates besides the 4- and 3-bit ones above (exl3_pstep.hip: PS_PAIRS) runs the step: against decode_step_fx on the same
    tensors (2e-2), without attention also against the oracle composition (3e-2); replay == eager; no time-out."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    shape = LlamaShape("small-k%d" % K, 1024, 2816, 2, 8, 2, 128, 3072)
    m = SyntheticEXL3Llama(shape, K=K, cb=2, device=dev, kv_bits=4, max_ctx=1024, head_K=head_K)
    m.alloc_state(1, pos=300)
    m.with_attention = with_attention
    assert m.lm_head.K == (head_K or K) and m.persistent_applies()
    if not with_attention:
        ref = _oracle_decode(m, _np(m.x0))
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    lf = _np(m.decode_step_fx().float()).copy()
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    lp = _np(m.decode_step_persistent().float()).copy()
    assert m._pstep is not None and not m._pstep.error() and ("head_K=%d" % (head_K or K)) in m._pstep.describe()
    assert np.isfinite(lp).all() and _relerr(lp, lf) < 2e-2, _relerr(lp, lf)
    if not with_attention:
        assert _relerr(lp, ref) < 3e-2
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    _replay_equals(m.decode_step_persistent, m, lp, reps=3)
    assert not m._pstep.error()


# shapes of other public Llama-shaped checkpoints: their plans use slice counts no BASELINE shape has (6, 7, 10, 12, 15, 16, 18, 25 slices per op on 256 CUs; slab sums of
# 12 / 16 lines, cooperative row gathers over 10 - 25 partial lines, uneven slices) -- two layers (the second q|k|v op takes the direct row edge) + a 16384-column head
MID_SHAPES = [("llama-3.2-3b", 3072, 8192, 24, 8, 128), ("qwen2.5-7b", 3584, 18944, 28, 4, 128), ("qwen2.5-1.5b", 1536, 8960, 12, 2, 128),
              ("llama-2-7b", 4096, 11008, 32, 32, 128), ("tinyllama", 2048, 5632, 32, 4, 64), ("hidden-2560", 2560, 6912, 20, 4, 128)]


@pytest.mark.parametrize("name,hidden,inter,hq,hkv,hd", MID_SHAPES)
def test_persistent_step_other_checkpoint_shapes_vs_launch_per_op(dev, name, hidden, inter, hq, hkv, hd):
    """Without the attention core: logits / finished queries / appended K / V rows of the persistent step against decode_step_fx on the same tensors (the launch-per-op
    GEMVs are shape-general and oracle-tested per shape in test_gpu_gemv / test_gpu_fullsize), eager twice == graph replay, no time-out."""
    from exllamav3_amd.llama_path import LlamaShape
    m = _model(LlamaShape(name + "-2layer", hidden, inter, 2, hq, hkv, hd, 16384), dev)
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    lf = _np(m.decode_step_fx().float()).copy()
    kv_f = [(_np(c).copy(), _np(s_).copy()) for c, s_ in m.kcache + m.vcache]
    q_f = _np(m.q.float()).copy()
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    m.q.zero_()
    lp = _np(m.decode_step_persistent().float()).copy()
    assert m._pstep is not None and not m._pstep.error()
    assert np.isfinite(lp).all()
    assert _relerr(lp, lf) < 2e-2, _relerr(lp, lf)
    assert np.abs(q_f - _np(m.q.float())).max() < 2e-2 * max(1.0, float(np.abs(q_f).max()))
    page, slot = int(m.block_table[0, 700 // m.page]), 700 % m.page
    for (wa, sa), (c, s_) in zip(kv_f, m.kcache + m.vcache):
        _kv_rows_match_in_levels(_np(c[page, slot]), _np(s_[page, slot]), wa[page, slot], sa[page, slot], 0.02)
        assert np.abs(_np(c)).sum() == np.abs(_np(c[page, slot])).sum()          # nothing but the new token's row was written
    assert np.array_equal(_np(m.decode_step_persistent().float()), lp)
    _replay_equals(m.decode_step_persistent, m, lp, reps=3)
    assert not m._pstep.error()


@pytest.mark.parametrize("name,hidden,inter,hq,hkv,hd", [s_ for s_ in MID_SHAPES if (s_[3] // s_[4]) * (128 // s_[5]) <= 8])
def test_persistent_step_other_checkpoint_shapes_with_attention_inside(dev, name, hidden, inter, hq, hkv, hd):
    from exllamav3_amd.llama_path import LlamaShape
    _attention_inside_case(dev, LlamaShape(name + "-2layer", hidden, inter, 2, hq, hkv, hd, 16384), 700)


# ---- round 6: the regimes the round-5 suite did not reach -----------------------------------------------------------------------------------------

def _kv_levels4(words):
    """4-bit cache words of one token row (G * 4 uint32: one width-4 plane per 32-group, element e at bit 4 e of the plane: cache/q_cache_kernels.cuh:130-147) -> levels (G, 32)."""
    w = np.ascontiguousarray(words).view(np.uint32).reshape(-1, 4)
    e = np.arange(32)
    return ((w[:, e // 8] >> ((e % 8) * 4).astype(np.uint32)) & np.uint32(15)).astype(np.int32)


def _kv_rows_match_in_levels(words_a, scales_a, words_b, scales_b, max_frac):
    """Two quantized rows of the same token from two pipelines whose inputs differ by rounding: EVERY level at most one step apart and at most `max_frac` of them different,
    every group scale within 2 fp16 ulps.  A wrong 32-group (slot, head, rope pairing, Hadamard) has most of its levels off by more than one step -- the whole-row RMS bound
    this replaces (0.2 sigma) let a single wrong group through."""
    la, lb = _kv_levels4(words_a), _kv_levels4(words_b)
    d = np.abs(la - lb)
    assert d.max() <= 1, f"levels up to {d.max()} steps apart (group {np.argwhere(d > 1)[0][0]})"
    assert (d != 0).mean() <= max_frac, f"{(d != 0).mean():.4f} of the levels differ"
    sa, sb = np.asarray(scales_a, dtype=np.float32).reshape(-1), np.asarray(scales_b, dtype=np.float32).reshape(-1)
    assert np.all(np.abs(sa - sb) <= 2.0 ** -9 * np.maximum(np.abs(sa), np.abs(sb)) + 1e-7)


@pytest.mark.parametrize("hd,pos,layers", [(128, 4096, 2), (128, 8191, 1), (128, 15999, 1), (64, 2048, 2), (64, 5999, 1)])
def test_persistent_step_attention_inside_multi_step_splits_vs_oracle(dev, hd, pos, layers):
    """The attention inside the persistent step where a context split takes MORE than one 128-token step (length > 32 x 128 at head_dim 128, > 16 x 128 at head_dim 64: the
    next-step prefetch and the lazy-softmax rescale across steps of exl3_pstep_kernel.cuh's token loop -- the regime of bench.py's ..._ctx16000 line, which no test reached:
    VERDICT r5 weak 1).  Llama-3.1-8B's / Llama-3.2-1B's attention geometry (32 / 8 heads), a random quantized context from the oracle's kv_quant, against
      * the ORACLE: attention over the dequantized cache (attn_decode_qcache), then the rest of the layer(s) and the head -- logits 3e-2, the appended K / V rows level by level;
      * decode_step_fx with the attention core on the same tensors (2e-2; rows level by level);
    nsteps > 1 asserted from the plan's geometry; eager twice == graph replay; no time-out."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    hidden = 4096 if hd == 128 else 2048
    shape = LlamaShape("att-long-hd%d" % hd, hidden, 2048, layers, 32, 8, hd, 4096)
    m = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=16384 if hd == 128 else 8192)
    m.alloc_state(1, pos=pos)
    m.with_attention = True
    m.attn_merge_in_oproj_hd64 = True
    assert m.persistent_applies()
    rng = np.random.default_rng(1000 * hd + pos)
    ctxs = [_fill_ctx(m, li, 1, rng, dev) for li in range(layers)]
    saved = [(c.clone(), s_.clone()) for c, s_ in m.kcache + m.vcache]

    def restore():
        for (c, s_), (c0, s0) in zip(m.kcache + m.vcache, saved):
            c.copy_(c0); s_.copy_(s0)
    # ---- oracle
    x, pend, kv_o = _np(m.x0), None, []
    for li, L in enumerate(m.layers):
        x, ov, k4, v = _oracle_attention_sublayer(m, L, x, pend, pos, tuple(a.copy() for a in ctxs[li]))
        kv_o.append((k4, v))
        xn, x = o.rms_norm(ov, _np(L["norm2"]), m.eps, residual_in=x)
        gf, uf = _lin(L["gate"], xn).astype(np.float32), _lin(L["up"], xn).astype(np.float32)
        pend = _lin(L["down"], (gf / (1 + np.exp(-gf)) * uf).astype(np.float16), out_fp32=True)
    xn_f, _ = o.rms_norm(pend, _np(m.final_norm), m.eps, residual_in=x)
    ref = _lin(m.lm_head, xn_f).astype(np.float32)
    # ---- the persistent step
    lp = _np(m.decode_step_persistent().float()).copy()
    assert m._pstep is not None and not m._pstep.error()
    nse, st_tok, nsteps = m._pstep.attn_geometry(pos + 1)
    assert nsteps > 1 and nse * st_tok >= pos + 1, (nse, st_tok, nsteps)
    assert np.isfinite(lp).all() and _relerr(lp, ref) < 3e-2, _relerr(lp, ref)
    page, slot = int(m.block_table[0, pos // m.page]), pos % m.page
    kv_p = [(_np(c[page, slot]).copy(), _np(s_[page, slot]).copy()) for c, s_ in m.kcache + m.vcache]
    for li in range(layers):
        for (wa, sa), ref_row in ((kv_p[li], kv_o[li][0]), (kv_p[layers + li], kv_o[li][1])):
            rq, rs = o.kv_quant(np.ascontiguousarray(ref_row).reshape(1, 1, -1).astype(np.float16), 4)
            _kv_rows_match_in_levels(wa, sa, rq.reshape(-1), rs.reshape(-1), 0.03)
    for (c, s_), (c0, _) in zip(m.kcache + m.vcache, saved):
        keep = np.ones(c.shape[:2], dtype=bool); keep[page, slot] = False
        assert np.array_equal(_np(c)[keep], _np(c0)[keep])                 # nothing but the new token's row was written
    # ---- the launch-per-op step with the attention core on the same tensors
    restore()
    lf = _np(m.decode_step_fx().float()).copy()
    assert _relerr(lp, lf) < 2e-2, _relerr(lp, lf)
    for (wa, sa), (c, s_) in zip(kv_p, m.kcache + m.vcache):
        _kv_rows_match_in_levels(wa, sa, _np(c[page, slot]), _np(s_[page, slot]), 0.02)
    restore()
    assert np.array_equal(_np(m.decode_step_persistent().float()), lp)
    restore()
    _replay_equals(m.decode_step_persistent, m, lp, reps=3)
    assert not m._pstep.error()


@pytest.mark.parametrize("K,head_K", [(4, None), (3, 6)])
def test_load_time_repack_is_a_permutation_and_changes_no_bit(dev, K, head_K):
    """exl3_pstep_create copies every op's packed words into the order the plan streams them (PsOp::Bp).  (1) exl3_pstep_unpack_op inverts the copy: every matrix of every op
    comes back EQUAL to the checkpoint tensor, word for word (same words, other addresses: F1's format is untouched); (2) the step over the repacked words gives the same
    logits, queries and cache rows -- bit for bit -- as the step that streams the checkpoint layout (repack=False)."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    shape = LlamaShape("repack", 1024, 2816, 2, 8, 2, 128, 3072)
    m = SyntheticEXL3Llama(shape, K=K, cb=2, device=dev, kv_bits=4, max_ctx=1024, head_K=head_K)
    m.alloc_state(1, pos=300)
    layers = [dict(L, kcache=m.kcache[i], vcache=m.vcache[i]) for i, L in enumerate(m.layers)]
    outs = {}
    for repack in (True, False):
        ps = ext.PersistentStep(layers, m.lm_head, m.final_norm, shape.hidden, m.hq, m.hkv, 128, m.eps, repack=repack)
        assert ("repacked" in ps.describe()) == repack
        if repack:
            for li, L in enumerate(m.layers):
                for g, names in enumerate((("q", "k", "v"), ("o",), ("gate", "up"), ("down",))):
                    for j, nm in enumerate(names):
                        assert torch.equal(ps.unpack_op(4 * li + g, j, L[nm].trellis), L[nm].trellis), (li, nm)
            assert torch.equal(ps.unpack_op(4 * len(m.layers), 0, m.lm_head.trellis), m.lm_head.trellis)
        else:
            with pytest.raises(RuntimeError):
                ps.unpack_op(0, 0, m.layers[0]["q"].trellis)
        for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
        m.q.zero_(); m.logits.zero_()
        ext.fx_init_prep(m.x0, m.R, m.ss, 1, m.inv_freq, m.positions, 128, m.block_table, m.page, m.rope_sin, m.rope_cos, m.kv_slots)
        ps.run(m.R, m.logits, m.q, m.rope_sin, m.rope_cos, m.kv_slots)
        assert not ps.error() and not ps.error_peek()
        outs[repack] = (_np(m.logits.float()).copy(), _np(m.q.float()).copy(), [(_np(c).copy(), _np(s_).copy()) for c, s_ in m.kcache + m.vcache])
        del ps
    assert np.isfinite(outs[True][0]).all() and np.abs(outs[True][0]).max() > 0
    assert np.array_equal(outs[True][0], outs[False][0]) and np.array_equal(outs[True][1], outs[False][1])
    for (wa, sa), (wb, sb) in zip(outs[True][2], outs[False][2]):
        assert np.array_equal(wa, wb) and np.array_equal(sa, sb)


def test_persistent_step_time_out_poisons_the_logits_and_drains_quickly(dev):
    """A step whose waits cannot be satisfied in the allowed polls (spin limit 1: the first tagged line that is not there yet times out) must (a) set the error word and its
    pinned host mirror, (b) write NaN logits -- never a plausible row -- and (c) come back at once: after the first time-out every later wait of the workgroup gives up after
    one poll (it used to restart the full limit ~129 x several times).  The next step with the normal limit is clean again."""
    import time
    from exllamav3_amd.llama_path import LlamaShape
    m = _model(LlamaShape("tiny-to", 512, 1536, 4, 8, 2, 64, 1024), dev)
    good = _np(m.decode_step_persistent().float()).copy()
    assert not m._pstep.error() and not m._pstep.error_peek()
    m._pstep.set(spin_limit=1)
    torch.cuda.synchronize(); t0 = time.time()
    from exllamav3_amd import ext
    # (the plan is run directly: decode_step_persistent would drop a plan that reports a time-out)
    ext.fx_init_prep(m.x0, m.R, m.ss, 1, m.inv_freq, m.positions, 64, m.block_table, m.page, m.rope_sin, m.rope_cos, m.kv_slots)
    m._pstep.run(m.R, m.logits, m.q, m.rope_sin, m.rope_cos, m.kv_slots)
    torch.cuda.synchronize(); dt = time.time() - t0
    bad = _np(m.logits.float())
    if m._pstep.error_peek():
        assert dt < 2.0, dt
        assert np.isnan(bad).any() and m._pstep.error()
    else:
        # (a chip fast enough that no line was ever late: nothing to poison)
        assert np.array_equal(bad, good)
    m._pstep.set(spin_limit=1 << 17)
    assert not m._pstep.error_peek()
    assert np.array_equal(_np(m.decode_step_persistent().float()), good)
    assert not m._pstep.error()


@pytest.mark.parametrize("K,head_K,with_attention", [(3, 6, False), (4, None, False), (3, 6, True)])
def test_persistent_step_rows_wider_than_4096_vs_oracle(dev, K, head_K, with_attention):
    """hidden 8192 (Llama-3.1-70B's width; round 6: the residual row may have 64 Hadamard blocks -- eight per service half-wave in the exact RMSNorm of the first op and the
    lm_head, two sum-of-squares granules per lane on the direct row edges, the lm_head's one 64-block slice of activation quads running into the gather area of the LDS map):
    two layers + a 4096-column head against the oracle composition (3e-2) and decode_step_fx (2e-2); replay == eager; no time-out."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    shape = LlamaShape("wide-8192", 8192, 3584, 2, 16, 2, 128, 4096)
    m = SyntheticEXL3Llama(shape, K=K, cb=2, device=dev, kv_bits=4, max_ctx=1024, head_K=head_K)
    m.alloc_state(1, pos=300)
    m.with_attention = with_attention
    assert m.persistent_applies()
    if not with_attention:
        ref = _oracle_decode(m, _np(m.x0))
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    lf = _np(m.decode_step_fx().float()).copy()
    kv_f = [(_np(c).copy(), _np(s_).copy()) for c, s_ in m.kcache + m.vcache]
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    lp = _np(m.decode_step_persistent().float()).copy()
    assert m._pstep is not None and not m._pstep.error()
    assert np.isfinite(lp).all() and _relerr(lp, lf) < 2e-2, _relerr(lp, lf)
    if not with_attention:
        assert _relerr(lp, ref) < 3e-2, _relerr(lp, ref)
    page, slot = int(m.block_table[0, 300 // m.page]), 300 % m.page
    for (wa, sa), (c, s_) in zip(kv_f, m.kcache + m.vcache):
        _kv_rows_match_in_levels(wa[page, slot], sa[page, slot], _np(c[page, slot]), _np(s_[page, slot]), 0.02)
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    _replay_equals(m.decode_step_persistent, m, lp, reps=3)
    assert not m._pstep.error()


def test_persistent_step_on_one_rank_of_a_70b_tp8_job_without_the_exchange(dev):
    """ONE rank's shards of Llama-3.1-70B at 3 bpw under TP = 8 (q 8192 -> 1024, k / v 8192 -> 128, o 1024 -> 8192, gate / up 8192 -> 3584, down 3584 -> 8192, a 16128-column
    head shard) with the collectives left out (tp.OneRankOfMany: bench.py's llama-3.1-70b_tp8_rank line measures this compute leg): the persistent step over the rank's
    tensors against the launch-per-op TP branch (decode_step_fx under TP, exchange left out the same way) -- logits 2e-2, appended rows level by level; replay == eager."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama
    from exllamav3_amd.tp import OneRankOfMany
    ext.set_gemv_variant(1)
    m = SyntheticEXL3Llama(SHAPES["llama-3.1-70b"], K=3, cb=2, device=dev, backend=OneRankOfMany(8, dev), kv_bits=4, max_ctx=1024, layers=3)
    m.alloc_state(1, pos=300)
    assert m.tp == 8 and m.hq == 8 and m.hkv == 1 and m.persistent_applies()
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    lf = _np(m.decode_step_fx().float()).copy()
    kv_f = [(_np(c).copy(), _np(s_).copy()) for c, s_ in m.kcache + m.vcache]
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    lp = _np(m.decode_step_persistent().float()).copy()
    assert m._pstep is not None and not m._pstep.error()
    assert np.isfinite(lp).all() and _relerr(lp, lf) < 2e-2, _relerr(lp, lf)
    page, slot = int(m.block_table[0, 300 // m.page]), 300 % m.page
    for (wa, sa), (c, s_) in zip(kv_f, m.kcache + m.vcache):
        _kv_rows_match_in_levels(wa[page, slot], sa[page, slot], _np(c[page, slot]), _np(s_[page, slot]), 0.02)
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    _replay_equals(m.decode_step_persistent, m, lp, reps=3)
    assert not m._pstep.error()
