"""HIP path vs the CPU oracle at the REAL shapes of BASELINE.json's configs: every GEMV of a Llama-3.1-70B TP=8 rank at
3 bpw, one fused decode layer at those shapes, lm_head 4096 -> 128256, a Mixtral-8x7B MoE block (4096 <-> 14336, 8 experts, top-2), the fused
decode step at batch 16, and one Llama-3.1-8B layer of the 4096-token prefill chunk on sampled token rows.

The numpy oracle decodes ~5 M weights per second per core, so the big matrices go through it in column chunks on a thread pool (numpy releases
the GIL) and lm_head is checked on column slices (oracle tp_slice == the reference's out-split shard, modules/quant/exl3.py:300-306)."""
import os
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import pytest
import torch
from oracle import exl3_oracle as o

pytestmark = pytest.mark.gpu

_POOL = ThreadPoolExecutor(max_workers=min(48, os.cpu_count() or 8))


def _np(t):
    return t.detach().cpu().numpy()


def _cb(L):
    return 1 if L.mcg else (2 if L.mul1 else 0)


def oracle_linear(x, tr, su, sv, K, cb, out_fp32=False, cols=None, chunk=512):
    """o.linear_forward over column ranges (default: all of n in `chunk`-column pieces), concatenated."""
    n = sv.shape[0]
    ranges = cols if cols is not None else [(c, min(c + chunk, n)) for c in range(0, n, chunk)]

    def one(r):
        t, s_u, s_v, _ = o.tp_slice(tr, su, sv, None, r[0], r[1], "n")
        return o.linear_forward(x, t, s_u, s_v, K, cb, out_fp32=out_fp32)
    return np.concatenate(list(_POOL.map(one, ranges)), axis=-1)


def _lin(L, x, out_fp32=False, cols=None):
    return oracle_linear(x, _np(L.trellis), _np(L.suh), _np(L.svh), L.K, _cb(L), out_fp32, cols)


def _relerr(got, ref):
    return float(np.abs(got.astype(np.float32) - ref.astype(np.float32)).max() / (np.sqrt((ref.astype(np.float32) ** 2).mean()) + 1e-12))


# Llama-3.1-70B at TP = 8, per-rank shards: q 8192->1024, k/v 8192->128, o 1024->8192, gate/up 8192->3584, down 3584->8192
SHAPES_70B_RANK = {"q": (8192, 1024), "k": (8192, 128), "v": (8192, 128), "o": (1024, 8192), "gate": (8192, 3584), "down": (3584, 8192)}


@pytest.mark.parametrize("name", list(SHAPES_70B_RANK))
@pytest.mark.parametrize("m", [1, 16])
def test_gemv_70b_tp8_rank_shapes_vs_oracle(dev, name, m):
    """exl3_gemm at K = 3 (config 4's bitrate), mul1, batch 1 (generation 2) and 16 (generation 3), fp16 and fp32 outputs."""
    from exllamav3_amd import ext
    k, n = SHAPES_70B_RANK[name]
    if m == 16 and name in ("k", "v", "gate"):
        pytest.skip("batch 16 on q / o / down covers generation 3 at these k and n")
    K, cb = 3, 2
    tr, su, sv = o.synth_linear(k, n, K, seed=k + n, realistic=True)
    x = np.random.default_rng(m).standard_normal((m, k)).astype(np.float16)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ttr, tsu, tsv, tx = T(tr), T(su), T(sv), T(x)
    fp32 = name in ("o", "down")                       # architecture/llama.py:95,111: o / down produce fp32
    y = torch.empty((m, n), dtype=torch.float if fp32 else torch.half, device=dev)
    ext.exl3_gemm(tx, ttr, y, tsu, torch.empty_like(tx), tsv, -1, False, True, 0)
    ref = oracle_linear(x, tr, su, sv, K, cb, out_fp32=fp32)
    assert _relerr(_np(y), ref) < 1e-2


def test_fused_decode_layer_at_70b_tp8_rank_shapes_vs_oracle(dev):
    """One fused decode layer (7 launches) whose linears have exactly a TP = 8 rank's shapes (hidden 8192, 8 q heads, 1 kv head, inter 3584), K = 3,
    against the oracle composition -- the shapes `test_fused_pipeline_on_70b_tp8_rank_shapes` only compared HIP-vs-HIP."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    from test_gpu_path import _oracle_decode
    import test_gpu_path
    ext.set_gemv_variant(1)
    shape = LlamaShape("70b-tp8-rank", 8192, 3584, 1, 8, 1, 128, 1024)
    model = SyntheticEXL3Llama(shape, K=3, cb=2, device=dev, kv_bits=4, max_ctx=1024)
    model.alloc_state(1, pos=300)
    logits = _np(model.decode_step_fused().float())
    saved = test_gpu_path._lin
    test_gpu_path._lin = _lin                           # the layer composition of test_gpu_path with the chunked / threaded linear
    try:
        ref = _oracle_decode(model, _np(model.x0))
    finally:
        test_gpu_path._lin = saved
    assert np.isfinite(logits).all()
    assert _relerr(logits, ref) < 3e-2


@pytest.mark.parametrize("m", [1, 2])
def test_lm_head_8b_vs_oracle_on_column_slices(dev, m):
    """lm_head 4096 -> 128256 (1002 column blocks) at 4 bpw: the whole GEMV on the GPU, the oracle on 10 column ranges of 128..256 columns incl. the
    first and the last block (out-split shard of the reference: svh[n0:n1], trellis[:, n0/16:n1/16])."""
    from exllamav3_amd import ext
    k, n, K = 4096, 128256, 4
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    from exllamav3_amd.llama_path import _rand_linear
    L = _rand_linear(k, n, K, 2, dev, gen)
    x = torch.randn((m, k), device=dev, generator=gen).half()
    y = torch.empty((m, n), dtype=torch.half, device=dev)
    ext.exl3_gemm(x, L.trellis, y, L.suh, torch.empty_like(x), L.svh, -1, False, True, 0)
    rng = np.random.default_rng(3)
    starts = sorted({0, n - 128, *(int(b) * 128 for b in rng.integers(1, 1000, size=8))})
    cols = [(s, min(s + 256, n)) if i % 2 else (s, s + 128) for i, s in enumerate(starts)]
    got = np.concatenate([_np(y)[:, a:b] for a, b in cols], axis=-1)
    tr, su, sv = _np(L.trellis), _np(L.suh), _np(L.svh)
    ref = oracle_linear(_np(x), tr, su, sv, K, 2, cols=cols)
    assert _relerr(got, ref) < 1e-2
    # the NORM-mode launch the fused step uses for the head (in-GEMV RMSNorm) on the same slices
    w = (1.0 + 0.05 * torch.randn(k, device=dev, generator=gen)).half()
    resid = (torch.randn((m, k), device=dev, generator=gen) * 3).half()
    ss = torch.zeros((m, k // 128), dtype=torch.float, device=dev)
    ext.glue_resid(None, 0, None, None, resid, ss, m)
    y2 = torch.empty_like(y)
    ext.exl3_gemv_ex_norm(resid, w, ss, 1e-5, [L.trellis], [y2], [L.suh], [L.svh], m, False, True, 0)
    xn = o.rms_norm(_np(resid), _np(w), 1e-5)
    ref2 = oracle_linear(xn, tr, su, sv, K, 2, cols=cols)
    got2 = np.concatenate([_np(y2)[:, a:b] for a, b in cols], axis=-1)
    assert _relerr(got2, ref2) < 1e-2


def test_moe_block_at_mixtral_shapes_vs_oracle(dev):
    """SyntheticEXL3MoE at Mixtral-8x7B's block shapes (hidden 4096, inter 14336, 8 experts, top-2, 4 bpw), bs 1: router, indexed gate|up launch,
    weighted down launch against the oracle over the two selected experts (config 5)."""
    from exllamav3_amd.moe_path import SyntheticEXL3MoE
    moe = SyntheticEXL3MoE(4096, 14336, experts=8, top_k=2, K=4, cb=2, device=dev, seed=11)
    x = torch.randn((1, 4096), device=dev, generator=torch.Generator(device=dev).manual_seed(5)).half()
    y = _np(moe.forward(x).float())
    xs = _np(x)
    scores, sel, w = o.routing_std(xs, _np(moe.router), 2)
    assert np.array_equal(_np(moe.sel), sel)
    assert np.abs(_np(moe.w.float()) - w.astype(np.float32)).max() < 2e-3
    ref = np.zeros((1, 4096), dtype=np.float32)
    for j in range(2):
        e = int(sel[0, j])
        g = _lin(moe.gate[e], xs).astype(np.float32); u = _lin(moe.up[e], xs).astype(np.float32)
        a = (g / (1 + np.exp(-g)) * u).astype(np.float16)
        ref += float(w[0, j]) * _lin(moe.down[e], a, out_fp32=True)
    assert _relerr(y, ref) < 2e-2


def test_prefill_chunk_8b_layer_vs_oracle_on_sampled_rows(dev):
    """The 4096-token prefill chunk through ONE Llama-3.1-8B layer (reconstruct_had_slice_t -> NT GEMM route, fused q|k|v and gate|up GEMMs,
    residual add in the GEMM epilogue) against the oracle on sampled token rows: with the attention core out of scope every row is independent
    through the linears, so the oracle runs 6 rows through the full-size layer.  Checks the residual stream after the layer (all hidden columns)
    and the last token's logits."""
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    shape = LlamaShape("8b-1layer", 4096, 14336, 1, 32, 8, 128, 1024)
    model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4)
    toks = 4096
    logits = _np(model.prefill_chunk(toks).float())
    assert np.isfinite(logits).all()
    rows = np.array([0, 1, 777, 2048, 4000, toks - 1])
    x = _np(model.px0)[rows]
    L = model.layers[0]
    inv = _np(model.inv_freq)
    xn = o.rms_norm(x, _np(L["norm1"]), model.eps)
    q, k = _lin(L["q"], xn), _lin(L["k"], xn)
    q4 = np.empty((len(rows), model.hq, 128), np.float16)
    for i, r in enumerate(rows):            # rope position = token index (prefill from position 0)
        qi, _ = o.rope(q[i].reshape(1, 1, model.hq, 128), k[i].reshape(1, 1, model.hkv, 128), inv, position=int(r), rope_mode=o.ROPE_NEOX)
        q4[i] = qi[0, 0]
    ov = _lin(L["o"], q4.reshape(len(rows), -1), out_fp32=True)
    xn, x = o.rms_norm(ov, _np(L["norm2"]), model.eps, residual_in=x)
    gf, uf = _lin(L["gate"], xn).astype(np.float32), _lin(L["up"], xn).astype(np.float32)
    a = (gf / (1 + np.exp(-gf)) * uf).astype(np.float16)
    d = _lin(L["down"], a, out_fp32=True)
    x = (x.astype(np.float32) + d).astype(np.float16)
    xl = o.rms_norm(x[-1:], _np(model.final_norm), model.eps)
    ref = _lin(model.lm_head, xl).astype(np.float32)
    assert _relerr(logits, ref) < 5e-2
    # residual stream rows after the layer (prefill_chunk keeps it for this check)
    got_x = _np(model.px_out)[rows].astype(np.float32)
    assert _relerr(got_x, x) < 2e-2


def test_prefill_attention_8b_chunk_vs_oracle_on_sampled_queries(dev):
    """exl3_attn_prefill_paged at BASELINE config 3's prefill size (4096 new tokens, Llama-3.1-8B heads 32 / 8 x 128, 256-token pages in a
    permuted block table): sampled query rows against the fp32 definition (softmax(q K^T / sqrt(d)) V over keys 0 .. i), plus the causal
    property at full size: changing the keys and values after position p leaves the outputs of queries <= p bit-identical."""
    from exllamav3_amd import ext
    T, hq, hkv, hd, page = 4096, 32, 8, 128, 256
    g = torch.Generator(device=dev); g.manual_seed(11)
    pages = T // page
    bt = torch.randperm(pages, device=dev, generator=g).to(torch.int32).view(1, pages)
    q = torch.randn((1, T, hq, hd), device=dev, generator=g).half()
    kp = torch.randn((pages, page, hkv, hd), device=dev, generator=g).half()
    vp = torch.randn((pages, page, hkv, hd), device=dev, generator=g).half()
    lens = torch.tensor([T], dtype=torch.int32, device=dev)
    out = torch.empty_like(q)
    ext.attn_prefill_paged(q, out, kp, vp, bt, lens)
    k = kp[bt[0].long()].reshape(T, hkv, hd).float().cpu().numpy(); v = vp[bt[0].long()].reshape(T, hkv, hd).float().cpu().numpy()
    qn, got = q[0].float().cpu().numpy(), out[0].float().cpu().numpy()
    rows = [0, 1, 63, 64, 255, 256, 1000, 2047, 2048, 3071, 4000, 4095]
    worst = 0.0
    for i in rows:
        for h in range(0, hq, 5):
            kk, vv = k[: i + 1, h // (hq // hkv)], v[: i + 1, h // (hq // hkv)]
            s = (kk @ qn[i, h]) * np.float32(hd ** -0.5)
            p = np.exp(s - s.max()); p /= p.sum()
            ref = p @ vv
            worst = max(worst, float(np.abs(got[i, h] - ref).max() / np.sqrt((ref ** 2).mean())))
    assert worst < 2e-2
    # causal property: perturb everything after position p
    p0 = 1500
    kp2, vp2 = kp.clone(), vp.clone()
    flat_k = kp2[bt[0].long()].reshape(T, hkv, hd); flat_v = vp2[bt[0].long()].reshape(T, hkv, hd)
    flat_k[p0 + 1:] = torch.randn_like(flat_k[p0 + 1:]); flat_v[p0 + 1:] = torch.randn_like(flat_v[p0 + 1:])
    kp2[bt[0].long()] = flat_k.view(pages, page, hkv, hd); vp2[bt[0].long()] = flat_v.view(pages, page, hkv, hd)
    out2 = torch.empty_like(q)
    ext.attn_prefill_paged(q, out2, kp2, vp2, bt, lens)
    assert torch.equal(out2[0, : p0 + 1], out[0, : p0 + 1]) and not torch.equal(out2[0, p0 + 1:], out[0, p0 + 1:])


# ---- the SHIPPED DEFAULT decode pipelines at their own shapes: bench.py's headline runs decode_step_fx (generation-4 NORMFX input,
# GEMV_OUT_ATOMIC epilogues at o_proj S = 8 / down k = 14336, fused q|k|v n = 6144 and gate|up n = 28672); these run exactly that, one layer of the real
# shape + the real lm_head, against the ORACLE (chunked / threaded oracle linear above), incl. hipGraph replay ----------------------------------------------

def _fill_ctx(model, li, bsz, rng, dev):
    """Random quantized context in layer li's cache (contiguous pages per sequence in these models); returns the oracle-side packed arrays."""
    hd, ctx = model.shape.head_dim, model.max_ctx
    ck = rng.standard_normal((bsz, ctx, model.hkv * hd)).astype(np.float16); cv = rng.standard_normal((bsz, ctx, model.hkv * hd)).astype(np.float16)
    kq, ks = o.kv_quant(ck, model.kv_bits); vq, vs = o.kv_quant(cv, model.kv_bits)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    kc, ksc = model.kcache[li]; vc, vsc = model.vcache[li]
    kc.copy_(T(kq.view(np.int32)).view(kc.shape)); ksc.copy_(T(ks).view(ksc.shape)); vc.copy_(T(vq.view(np.int32)).view(vc.shape)); vsc.copy_(T(vs).view(vsc.shape))
    return kq, ks, vq, vs


def _oracle_attention_sublayer(model, L, x, pend, pos, ctx=None):
    """norm1 (+ pending residual) -> q|k|v -> rope -> [quantized append + attention over the dequantized cache] -> o_proj.  Returns (x, o_out fp32, k4, v)."""
    s, b = model.shape, x.shape[0]
    hd = s.head_dim
    if pend is None: xn = o.rms_norm(x, _np(L["norm1"]), model.eps)
    else: xn, x = o.rms_norm(pend, _np(L["norm1"]), model.eps, residual_in=x)
    q, k, v = _lin(L["q"], xn), _lin(L["k"], xn), _lin(L["v"], xn)
    q4, k4 = o.rope(q.reshape(b, 1, model.hq, hd), k.reshape(b, 1, model.hkv, hd), _np(model.inv_freq), positions=_np(model.positions), rope_mode=o.ROPE_NEOX)
    att = q4.reshape(b, -1)
    if ctx is not None:
        kq, ks, vq, vs = ctx
        knq, kns = o.kv_quant(k4.reshape(b, 1, -1), model.kv_bits); vnq, vns = o.kv_quant(v.reshape(b, 1, -1), model.kv_bits)
        kq[:, pos:pos + 1] = knq; ks[:, pos:pos + 1] = kns; vq[:, pos:pos + 1] = vnq; vs[:, pos:pos + 1] = vns
        kd = o.kv_dequant(kq, ks, model.kv_bits).reshape(b, -1, model.hkv, hd); vd = o.kv_dequant(vq, vs, model.kv_bits).reshape(b, -1, model.hkv, hd)
        att = o.attn_decode_qcache(q4.reshape(b, model.hq, hd), kd, vd, [pos + 1] * b).reshape(b, -1)
    return x, _lin(L["o"], att, out_fp32=True), k4, v


def _head_cols(vocab, seed=3):
    rng = np.random.default_rng(seed)
    starts = sorted({0, vocab - 128, *(int(b) * 128 for b in rng.integers(1, vocab // 128 - 1, size=6))})
    return [(s0, min(s0 + 256, vocab)) if i % 2 else (s0, s0 + 128) for i, s0 in enumerate(starts)]


def _check_logits_on_cols(logits, model, xn_final, cols, tol=3e-2):
    ref = _lin(model.lm_head, xn_final, cols=cols).astype(np.float32)
    got = np.concatenate([logits[:, a:b] for a, b in cols], axis=-1)
    assert np.isfinite(logits).all()
    err = _relerr(got, ref)
    assert err < tol, err
    return ref


def _replay_equals(step, model, logits, reps=3):
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            step()
    for _ in range(reps):
        model.logits.zero_(); g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(_np(model.logits.float()), logits)


def _appended_kv_close(model, li, bsz, pos, k4, v, tol=0.2):
    """The K / V rows the step appended to the quantized cache against the oracle's quantize -> dequantize of ITS rope(k) / v.  Both sides are dequantized
    with the oracle; the inputs differ by the linears' rounding (1e-3), so a few levels may flip (one flip moves the 32 values of its group by
    step / sqrt(32) ~ 0.05 sigma at 4 bits): bound 0.2 of the row's RMS, which a wrong slot / head / rope pairing misses by an order of magnitude."""
    kc, ks = model.kcache[li]; vc, vs = model.vcache[li]
    bits = model.kv_bits
    for b in range(bsz):
        page, slot = int(model.block_table[b, pos // model.page]), pos % model.page
        for (c, s_), ref in (((kc, ks), k4[b].reshape(1, 1, -1)), ((vc, vs), v[b].reshape(1, 1, -1))):
            got = o.kv_dequant(_np(c[page, slot]).view(np.uint32)[None, None], _np(s_[page, slot])[None, None], bits).reshape(-1).astype(np.float32)
            rq, rs = o.kv_quant(np.ascontiguousarray(ref).astype(np.float16), bits)
            want = o.kv_dequant(rq, rs, bits).reshape(-1).astype(np.float32)
            assert np.abs(got - want).max() / np.sqrt((want ** 2).mean()) < tol * (16.0 / 2 ** bits)


@pytest.mark.parametrize("bsz,cb,with_attention", [(1, 2, False), (2, 2, False), (1, 0, False), (1, 2, True)])
def test_fx_decode_step_llama_8b_layer_and_head_vs_oracle(dev, bsz, cb, with_attention):
    """decode_step_fx -- bench.py's headline pipeline -- on ONE Llama-3.1-8B layer (hidden 4096, 32 / 8 heads x 128, inter 14336) + the 128256-column
    lm_head at 4 bpw, batch 1 and 2, mul1 and 3INST, both forms of the MLP (glue_act_rs launch / silu * mul inside the down launch), and with the
    decode attention over a 1000-token 4-bit cache in the step (the driver line's with_attention config): logits on sampled column ranges, the final
    residual (all hidden columns) and the appended K / V rows against the oracle; hipGraph replay reproduces the eager bits."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    shape = LlamaShape("8b-1layer", 4096, 14336, 1, 32, 8, 128, 128256)
    model = SyntheticEXL3Llama(shape, K=4, cb=cb, device=dev, kv_bits=4, max_ctx=1024)
    pos = 1000 if with_attention else 700
    model.alloc_state(bsz, pos=pos)
    model.with_attention = with_attention
    assert bsz <= model.fx_max_bsz                         # i.e. bench.py really takes this pipeline at this batch
    ctx = _fill_ctx(model, 0, bsz, np.random.default_rng(5), dev) if with_attention else None
    # ---- oracle, once
    L = model.layers[0]
    x, ov, k4, v = _oracle_attention_sublayer(model, L, _np(model.x0), None, pos, ctx)
    xn, x = o.rms_norm(ov, _np(L["norm2"]), model.eps, residual_in=x)
    gf, uf = _lin(L["gate"], xn).astype(np.float32), _lin(L["up"], xn).astype(np.float32)
    a = (gf / (1 + np.exp(-gf)) * uf).astype(np.float16)
    d = _lin(L["down"], a, out_fp32=True)
    xn_f, x_f = o.rms_norm(d, _np(model.final_norm), model.eps, residual_in=x)
    cols = _head_cols(shape.vocab)
    for act_in in (False, True):
        model.fx_act_in_gemv = act_in
        for c, s_ in model.kcache + model.vcache:
            if not with_attention: c.zero_(); s_.zero_()
        logits = _np(model.decode_step_fx().float()).copy()
        _check_logits_on_cols(logits, model, xn_f, cols)
        assert _relerr(_np(model.x_final.float()), x_f) < 1e-2
        _appended_kv_close(model, 0, bsz, pos, k4, v)
        assert np.array_equal(_np(model.decode_step_fx().float()), logits)          # integer atomics: the same bits every time
        _replay_equals(model.decode_step_fx, model, logits)


def test_fx_decode_step_llama_1b_layer_and_head_vs_oracle(dev):
    """Llama-3.2-1B's default step (config 2: hidden 2048, head_dim 64, inter 8192; decode_step_fx in its 5-launch form, silu * mul inside the down
    launch) on one layer + the lm_head against the oracle, then graph replay."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    shape = LlamaShape("1b-1layer", 2048, 8192, 1, 32, 8, 64, 128256)
    model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=1024)
    assert model.fx_act_in_gemv                             # the default of this size
    model.alloc_state(1, pos=700)
    L = model.layers[0]
    x, ov, k4, v = _oracle_attention_sublayer(model, L, _np(model.x0), None, 700)
    xn, x = o.rms_norm(ov, _np(L["norm2"]), model.eps, residual_in=x)
    gf, uf = _lin(L["gate"], xn).astype(np.float32), _lin(L["up"], xn).astype(np.float32)
    a = (gf / (1 + np.exp(-gf)) * uf).astype(np.float16)
    d = _lin(L["down"], a, out_fp32=True)
    xn_f, x_f = o.rms_norm(d, _np(model.final_norm), model.eps, residual_in=x)
    logits = _np(model.decode_step_fx().float()).copy()
    _check_logits_on_cols(logits, model, xn_f, _head_cols(shape.vocab))
    assert _relerr(_np(model.x_final.float()), x_f) < 1e-2
    _appended_kv_close(model, 0, 1, 700, k4, v)
    _replay_equals(model.decode_step_fx, model, logits)


@pytest.mark.parametrize("with_attention", [False, True])
def test_fx_decode_step_mixtral_8x7b_layer_vs_oracle(dev, with_attention):
    """mixtral_path.decode_step_fx (config 5's default on one rank: MoE block in 3 launches, weighted expert rows added into the fixed-point residual)
    on ONE Mixtral-8x7B layer (hidden 4096, inter 14336, 8 experts top-2, vocab 32000) against the ORACLE composition -- not the launch-per-op step."""
    from exllamav3_amd.mixtral_path import MixtralShape, SyntheticEXL3Mixtral
    shape = MixtralShape("mixtral-1layer", 4096, 14336, 1, 32, 8, 128, 32000, 8, 2)
    model = SyntheticEXL3Mixtral(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=1024)
    pos = 1000 if with_attention else 300
    model.alloc_state(1, pos=pos)
    model.with_attention = with_attention
    ctx = _fill_ctx(model, 0, 1, np.random.default_rng(9), dev) if with_attention else None
    logits = _np(model.decode_step_fx().float()).copy()
    L = model.layers[0]; moe = L["moe"]
    x, ov, k4, v = _oracle_attention_sublayer(model, L, _np(model.x0), None, pos, ctx)
    xn, x = o.rms_norm(ov, _np(L["norm2"]), model.eps, residual_in=x)
    _, sel, w = o.routing_std(xn, _np(moe.router), 2)
    assert np.array_equal(_np(moe.sel), sel)
    y = np.zeros((1, shape.hidden), dtype=np.float32)
    for j in range(2):
        e = int(sel[0, j])
        g = _lin(moe.gate[e], xn).astype(np.float32); u = _lin(moe.up[e], xn).astype(np.float32)
        a = (g / (1 + np.exp(-g)) * u).astype(np.float16)
        y += float(w[0, j]) * _lin(moe.down[e], a, out_fp32=True)
    xn_f, x_f = o.rms_norm(y, _np(model.final_norm), model.eps, residual_in=x)
    ref = _lin(model.lm_head, xn_f).astype(np.float32)
    assert np.isfinite(logits).all()
    assert _relerr(logits, ref) < 3e-2
    assert _relerr(_np(model.x.float()), x_f) < 1e-2
    _appended_kv_close(model, 0, 1, pos, k4, v)
    _replay_equals(model.decode_step_fx, model, logits)


def test_moe_prefill_block_at_mixtral_shapes_vs_oracle_on_sampled_rows(dev):
    """SyntheticEXL3MoE.forward_prefill at config 5's block shapes (hidden 4096, inter 14336, 8 experts, top-2, 4 bpw) over a 4096-token chunk: ~1024 rows
    per expert through reconstruct_had + the NT GEMM.  The oracle runs 6 sampled token rows (a token's output depends on its own row only): routing
    of those rows identical, outputs to 2e-2 of their RMS."""
    from exllamav3_amd.moe_path import SyntheticEXL3MoE
    moe = SyntheticEXL3MoE(4096, 14336, experts=8, top_k=2, K=4, cb=2, device=dev, seed=11)
    T = 4096
    x = torch.randn((T, 4096), device=dev, generator=torch.Generator(device=dev).manual_seed(5)).half()
    y = _np(moe.forward_prefill(x))
    rows = np.array([0, 1, 777, 2048, 4000, T - 1])
    xs = _np(x)[rows]
    _, sel, w = o.routing_std(xs, _np(moe.router), 2)
    assert np.array_equal(_np(moe.pf_sel)[rows], sel)
    ref = np.zeros((len(rows), 4096), dtype=np.float32)
    for t in range(len(rows)):
        for j in range(2):
            e = int(sel[t, j])
            g = _lin(moe.gate[e], xs[t:t + 1]).astype(np.float32); u = _lin(moe.up[e], xs[t:t + 1]).astype(np.float32)
            a = (g / (1 + np.exp(-g)) * u).astype(np.float16)
            ref[t] += float(w[t, j]) * _lin(moe.down[e], a, out_fp32=True)[0]
    assert np.isfinite(y).all()
    assert _relerr(y[rows], ref) < 2e-2
