"""End-to-end parity of one decode step / one prefill chunk of the Llama-shaped hot path (exllamav3_amd/llama_path.py)
against a numpy composition of the oracle ops on the same synthetic tensors."""
import numpy as np
import pytest
import torch
from oracle import exl3_oracle as o

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _lin(L, x, out_fp32=False):
    cb = 1 if L.mcg else (2 if L.mul1 else 0)
    return o.linear_forward(x, _np(L.trellis), _np(L.suh), _np(L.svh), L.K, cb, out_fp32=out_fp32)


def _oracle_decode(model, x0):
    s = model.shape
    x = x0.copy()
    pending = None
    for L in model.layers:
        if pending is None:
            xn = o.rms_norm(x, _np(L["norm1"]), model.eps)
        else:
            xn, x = o.rms_norm(pending, _np(L["norm1"]), model.eps, residual_in=x)
        q, k, v = _lin(L["q"], xn), _lin(L["k"], xn), _lin(L["v"], xn)
        b = x.shape[0]
        pos = _np(model.positions)
        q4, k4 = o.rope(q.reshape(b, 1, model.hq, s.head_dim), k.reshape(b, 1, model.hkv, s.head_dim), _np(model.inv_freq),
                        positions=pos, rope_mode=o.ROPE_NEOX)
        ov = _lin(L["o"], q4.reshape(b, -1), out_fp32=True)
        xn, x = o.rms_norm(ov, _np(L["norm2"]), model.eps, residual_in=x)
        g, u = _lin(L["gate"], xn), _lin(L["up"], xn)
        gf, uf = g.astype(np.float32), u.astype(np.float32)
        a = (gf / (1 + np.exp(-gf)) * uf).astype(np.float16)
        pending = _lin(L["down"], a, out_fp32=True)
    xn, x = o.rms_norm(pending, _np(model.final_norm), model.eps, residual_in=x)
    return _lin(model.lm_head, xn).astype(np.float32)


@pytest.mark.parametrize("cb", [0, 2])
@pytest.mark.parametrize("bsz", [1, 3])
def test_decode_step_matches_oracle(dev, cb, bsz):
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    shape = LlamaShape("tiny", 256, 512, 2, 4, 2, 128, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=cb, device=dev, kv_bits=4, max_ctx=2048)
    model.alloc_state(bsz, pos=700)
    logits = model.decode_step().float().cpu().numpy()
    ref = _oracle_decode(model, _np(model.x0))
    assert np.isfinite(logits).all()
    err = np.abs(logits - ref).max() / np.sqrt((ref ** 2).mean())
    assert err < 3e-2, err
    # idempotent + graph replay gives the same bits as the eager step
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            model.decode_step()
    model.logits.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(model.logits.float().cpu().numpy(), logits)
    # the quantized KV append landed in the page / slot the block table names
    kc, ks = model.kcache[0]
    ksf = ks.float()
    assert float(ksf[model.block_table[0, 700 // 256].item(), 700 % 256].abs().sum()) > 0
    assert float(ksf.abs().sum()) == pytest.approx(float(sum(ksf[model.block_table[b, 700 // 256].item(), 700 % 256].abs().sum() for b in range(bsz))), rel=1e-5)


def test_prefill_chunk_runs_and_matches_small_oracle(dev):
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    shape = LlamaShape("tiny", 256, 512, 1, 4, 2, 128, 256)
    model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=8)
    toks = 1100                                  # >= 1024 rows: fused reconstruct_had + hgemm path
    logits = model.prefill_chunk(toks).float().cpu().numpy()
    assert logits.shape == (1, 256) and np.isfinite(logits).all()
    # oracle for the last token only needs every row through layer 0's linears -> do the full (small) computation
    x = _np(model.px0)
    L = model.layers[0]
    xn = o.rms_norm(x, _np(L["norm1"]), model.eps)
    q, k = _lin(L["q"], xn), _lin(L["k"], xn)
    q4, _ = o.rope(q.reshape(1, toks, model.hq, 128), k.reshape(1, toks, model.hkv, 128), _np(model.inv_freq), position=0, rope_mode=o.ROPE_NEOX)
    ov = _lin(L["o"], q4.reshape(toks, -1), out_fp32=True)
    xn, x = o.rms_norm(ov, _np(L["norm2"]), model.eps, residual_in=x)
    gf, uf = _lin(L["gate"], xn).astype(np.float32), _lin(L["up"], xn).astype(np.float32)
    a = (gf / (1 + np.exp(-gf)) * uf).astype(np.float16)
    d = _lin(L["down"], a, out_fp32=True)
    x = (x.astype(np.float32) + d).astype(np.float16)
    xl = o.rms_norm(x[-1:], _np(model.final_norm), model.eps)
    ref = _lin(model.lm_head, xl).astype(np.float32)
    err = np.abs(logits - ref).max() / np.sqrt((ref ** 2).mean())
    assert err < 5e-2, err


@pytest.mark.parametrize("cb", [0, 1, 2])
@pytest.mark.parametrize("bsz", [1, 2, 5, 16])
def test_fused_decode_pipeline_matches_unfused_and_oracle(dev, cb, bsz):
    """Deferred-epilogue GEMVs + glue kernels (exl3_glue.hip) against the op-by-op pipeline and the oracle."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    shape = LlamaShape("tiny", 256, 512, 2, 4, 2, 128, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=cb, device=dev, kv_bits=4, max_ctx=2048)
    model.alloc_state(bsz, pos=700)
    lu = model.decode_step().float().cpu().numpy().copy()
    ku = [(kc.clone(), ks.clone()) for kc, ks in model.kcache]
    vu = [(vc.clone(), vs.clone()) for vc, vs in model.vcache]
    for kc, ks in model.kcache + model.vcache:
        kc.zero_(); ks.zero_()
    lf = model.decode_step_fused().float().cpu().numpy()
    assert np.isfinite(lf).all()
    rms = np.sqrt((lu ** 2).mean())
    assert np.abs(lf - lu).max() / rms < 1e-2
    ref = _oracle_decode(model, _np(model.x0))                                  # every batch size, incl. the generation-3 route at 5 and 16
    assert np.abs(lf - ref).max() / np.sqrt((ref ** 2).mean()) < 3e-2
    # quantized KV append: same slots, same values up to fp32 summation order
    for (kc, ks), (kc0, ks0) in zip(model.kcache + model.vcache, ku + vu):
        assert bool(((ks != 0) == (ks0 != 0)).all())
        d1 = torch.empty((kc.shape[0] * kc.shape[1], kc.shape[2] // 4 * 32), dtype=torch.half, device=dev); d0 = torch.empty_like(d1)
        ext.dequant_cache_cont(kc.view(d1.shape[0], -1), ks.view(d1.shape[0], -1), d1)
        ext.dequant_cache_cont(kc0.view(d1.shape[0], -1), ks0.view(d1.shape[0], -1), d0)
        assert float((d1.float() - d0.float()).abs().max()) < 0.35          # one 4-bit quantization step at most
        assert float((d1.float() - d0.float()).abs().mean()) < 2e-3
    # graph replay reproduces the eager bits
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            model.decode_step_fused()
    model.logits.zero_(); g.replay(); torch.cuda.synchronize()
    assert np.array_equal(model.logits.float().cpu().numpy(), lf)
    # the bench's roofline leg launches every GEMV of the step on its own (bench.py): they must run with the rank's shapes too
    calls = model.gemv_calls("glue")
    assert len(calls) == 4 * model.n_layers + 1
    for c in calls:
        c()
    torch.cuda.synchronize()
    assert sum(cnt for (_, _, cnt) in model.gemv_launches_per_step()) == len(calls)


def test_fused_gptj_rope_and_gemv_ex_modes(dev):
    """exl3_gemv_ex: rotated input + final output equals exl3_gemm on the unrotated input; glue_qkv GPTJ rope vs rope kernel."""
    from exllamav3_amd import ext
    k, n, K, m = 512, 256, 4, 3
    tr, suh, svh = o.synth_linear(k, n, K, realistic=True)
    x = np.random.default_rng(0).standard_normal((m, k)).astype(np.float16)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tx, ttr, tsu, tsv = T(x), T(tr), T(suh), T(svh)
    for cb in (0, 2):
        y0 = torch.empty((m, n), dtype=torch.half, device=dev)
        ext.exl3_gemm(tx, ttr, y0, tsu, None, tsv, -1, False, cb == 2, 0)
        xh = torch.empty_like(tx); ext.had_r_128(tx, xh, tsu, None, 1.0)
        xs = xh.float().view(m, k // 128, 128).sum(-1).contiguous()
        y1 = torch.full((m, n), float("nan"), dtype=torch.half, device=dev)
        ext.exl3_gemv_ex(None, [xh], [xs], [ttr], [y1], None, [tsv], m, False, cb == 2, ext.GEMV_IN_ROTATED)
        assert float((y1.float() - y0.float()).abs().max()) < 2e-3 * float(y0.float().abs().max()) + 1e-3
    # glue_qkv with GPTJ rope against the standalone ops
    hq, hkv = 2, 1
    mats = [o.synth_linear(k, h * 128, K, seed=9 + i, realistic=True) for i, h in enumerate((hq, hkv, hkv))]
    Bs = [T(t[0]) for t in mats]; su = [T(t[1]) for t in mats]; sv = [T(t[2]) for t in mats]
    outs = [torch.empty((m, h * 128), dtype=torch.half, device=dev) for h in (hq, hkv, hkv)]
    ext.exl3_mgemm_bcast(tx, Bs, outs, su, sv, False, True)
    inv = T((1.0 / (10000.0 ** (np.arange(0, 128, 2) / 128))).astype(np.float32))
    pos = T(np.array([5, 900, 31000], dtype=np.int32))
    q4, k4 = outs[0].view(m, 1, hq, 128).clone(), outs[1].view(m, 1, hkv, 128).clone()
    ext.rope(q4, q4, k4, k4, inv, 0, pos, None, 1, 1.0)
    slabs, S = ext.exl3_gemv_ex(tx, None, None, Bs, None, su, None, m, False, True, ext.GEMV_OUT_DEFERRED)
    qf = torch.empty((m, hq * 128), dtype=torch.half, device=dev); kf = torch.empty((m, hkv * 128), dtype=torch.half, device=dev)
    vf = torch.empty((m, hkv * 128), dtype=torch.half, device=dev)
    ext.glue_qkv(slabs, S, sv[0], sv[1], sv[2], qf, kf, vf, inv, pos, None, None, None, None, None, 256, 8, 8, m, hq, hkv, 128, rope_mode=1)
    for a, b in ((qf, q4.view(m, -1)), (kf, k4.view(m, -1)), (vf, outs[2])):
        assert float((a.float() - b.float()).abs().max()) < 4e-3 * float(b.float().abs().max()) + 2e-3


@pytest.mark.parametrize("cb", [0, 1, 2])
@pytest.mark.parametrize("bsz", [1, 2, 5, 16])
def test_tail_epilogue_pipeline_is_bit_identical_to_glue_pipeline(dev, cb, bsz):
    """In-kernel tail epilogues (exl3_gemv2_tail.cuh, 4 launches/layer) run the glue kernels' arithmetic on the same slabs in the
    same summation order: logits, residual stream, q and the quantized KV pages must match the glue pipeline bit for bit."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    # the tail epilogues live in the generation-2 kernel: the glue side is pinned to it too (generations 3 / 4 group the partial sums differently)
    ext.set_gemm3_min_rows(0)
    ext.set_gemv_gen4(False)
    shape = LlamaShape("tiny", 256, 512, 3, 4, 2, 128, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=cb, device=dev, kv_bits=4, max_ctx=2048)
    model.alloc_state(bsz, pos=1234)
    lg = model.decode_step_fused_v1().clone()
    xg, qg = model.x.clone(), model.q.clone()
    kvg = [(c.clone(), s.clone()) for c, s in model.kcache + model.vcache]
    for c, s in model.kcache + model.vcache:
        c.zero_(); s.zero_()
    model.q.zero_(); model.logits.zero_()
    for rep in range(3):                                   # tickets must be re-armed: repeated launches give the same bits
        lt = model.decode_step_tail()
        torch.cuda.synchronize()
        assert torch.equal(lt, lg), f"logits differ (rep {rep})"
        assert torch.equal(model.x, xg) and torch.equal(model.q, qg)
        for (c, s), (c0, s0) in zip(model.kcache + model.vcache, kvg):
            assert torch.equal(c, c0) and torch.equal(s, s0)
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            model.decode_step_tail()
    model.logits.zero_(); g.replay(); g.replay(); torch.cuda.synchronize()
    assert torch.equal(model.logits, lg)


@pytest.mark.parametrize("K", [2, 3, 5, 8])
def test_tail_epilogue_entry_points_raw_input_gptj_and_bits(dev, K):
    """exl3_gemv_qkv / _act / _norm with un-rotated inputs (A + suh), GPTJ rope, 8-bit K / 3-bit V cache, bias, xn_out, a_out:
    against deferred GEMV + glue kernels (bit-exact) for every bitrate."""
    from exllamav3_amd import ext
    ext.set_gemv_gen4(False)                                # the tail entry points are generation-2 launches: bit-exactness needs the same kernel on the other side
    rng = np.random.default_rng(K)
    k, m, hq, hkv, inter = 512, 3, 3, 1, 384
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    x = T(rng.standard_normal((m, k)).astype(np.float16))
    mats = [o.synth_linear(k, h * 128, K, seed=20 + i, realistic=True) for i, h in enumerate((hq, hkv, hkv))]
    Bs = [T(t[0]) for t in mats]; su = [T(t[1]) for t in mats]; sv = [T(t[2]) for t in mats]
    inv = T((1.0 / (10000.0 ** (np.arange(0, 128, 2) / 128))).astype(np.float32))
    pos = T(np.array([7, 300, 1999], dtype=np.int32))
    page, pages = 256, 8
    bt = T(np.array([[3, 1, 0, 2, 7, 6, 5, 4], [0, 1, 2, 3, 4, 5, 6, 7], [7, 6, 5, 4, 3, 2, 1, 0]], dtype=np.int32))[:, :pages].contiguous()
    def caches(bits):
        return (torch.zeros((pages * 3, page, hkv * 128 // 32 * bits), dtype=torch.int32, device=dev),
                torch.zeros((pages * 3, page, hkv * 128 // 32), dtype=torch.half, device=dev))
    for mode in (1, 2):
        kc0, ks0 = caches(8); vc0, vs0 = caches(3); kc1, ks1 = caches(8); vc1, vs1 = caches(3)
        q0 = torch.empty((m, hq * 128), dtype=torch.half, device=dev); k0 = torch.empty((m, hkv * 128), dtype=torch.half, device=dev); v0 = torch.empty_like(k0)
        q1, k1, v1 = torch.empty_like(q0), torch.empty_like(k0), torch.empty_like(v0)
        slabs, S = ext.exl3_gemv_ex(x, None, None, Bs, None, su, None, m, False, False, ext.GEMV_OUT_DEFERRED)
        ext.glue_qkv(slabs, S, sv[0], sv[1], sv[2], q0, k0, v0, inv, pos, kc0, ks0, vc0, vs0, bt, page, 8, 3, m, hq, hkv, 128, rope_mode=mode)
        sn = torch.empty((m, 64), dtype=torch.float32, device=dev); cs = torch.empty_like(sn)
        ext.rope_table(inv, pos, sn, cs)
        ext.exl3_gemv_qkv(x, None, None, Bs, su, sv, m, False, False, q1, k1, v1, sn, cs, pos, kc1, ks1, vc1, vs1, bt, page, 8, 3, hq, hkv, 128, rope_mode=mode)
        for a, b in ((q0, q1), (k0, k1), (v0, v1), (kc0, kc1), (ks0, ks1), (vc0, vc1), (vs0, vs1)):
            assert torch.equal(a, b)
        assert int((ks1 != 0).sum()) == m * hkv * 4
    # gate/up + act
    gu = [o.synth_linear(k, inter, K, seed=40 + i, realistic=True) for i in range(2)]
    dn = o.synth_linear(inter, k, K, seed=50, realistic=True)
    Bg = [T(t[0]) for t in gu]; sug = [T(t[1]) for t in gu]; svg = [T(t[2]) for t in gu]
    Bd, sud, svd = T(dn[0]), T(dn[1]), T(dn[2])
    xh0 = torch.empty((m, inter), dtype=torch.half, device=dev); xs0 = torch.empty((m, inter // 128), dtype=torch.float32, device=dev); a0 = torch.empty_like(xh0)
    xh1, xs1, a1 = torch.empty_like(xh0), torch.empty_like(xs0), torch.empty_like(a0)
    slabs, S = ext.exl3_gemv_ex(x, None, None, Bg, None, sug, None, m, False, False, ext.GEMV_OUT_DEFERRED)
    ext.glue_act(slabs, S, svg[0], svg[1], sud, xh0, xs0, m, a_out=a0)
    ext.exl3_gemv_act(x, None, None, Bg, sug, svg, m, False, False, sud, xh1, xs1, a_out=a1)
    assert torch.equal(xh0, xh1) and torch.equal(xs0, xs1) and torch.equal(a0, a1)
    # down + bias + residual + norm + 2 consumers + xn_out
    bias = T((rng.standard_normal(k) * 0.1).astype(np.float16))
    w = T((1 + 0.1 * rng.standard_normal(k)).astype(np.float16))
    r0 = T(rng.standard_normal((m, k)).astype(np.float16)); r1 = r0.clone()
    t0 = [torch.empty((m, k), dtype=torch.half, device=dev) for _ in range(2)]; s0 = [torch.empty((m, k // 128), dtype=torch.float32, device=dev) for _ in range(2)]
    t1 = [torch.empty_like(t) for t in t0]; s1 = [torch.empty_like(t) for t in s0]
    xn0 = torch.empty((m, k), dtype=torch.half, device=dev); xn1 = torch.empty_like(xn0)
    slabs, S = ext.exl3_gemv_ex(None, [xh0], [xs0], [Bd], None, None, None, m, False, False, ext.GEMV_IN_ROTATED | ext.GEMV_OUT_DEFERRED)
    ext.glue_norm(slabs[0], S, svd, bias, r0, w, 1e-5, su[:2], t0, s0, m, xn_out=xn0)
    ext.exl3_gemv_norm(None, xh0, xs0, Bd, None, svd, bias, m, False, False, r1, w, 1e-5, su[:2], t1, s1, xn_out=xn1)
    assert torch.equal(r0, r1) and torch.equal(xn0, xn1)
    for a, b in zip(t0 + s0, t1 + s1):
        assert torch.equal(a, b)



@pytest.mark.parametrize("cb", [0, 1, 2])
@pytest.mark.parametrize("bsz", [1, 3, 16])
def test_in_gemv_rmsnorm_pipeline_is_bit_identical_to_single_workgroup_norm(dev, cb, bsz):
    """glue_resid + GEMV_IN_NORM (RMSNorm finished inside the consumer GEMV) against glue_norm (one workgroup normalises the row and
    rotates it for every consumer): same arithmetic and summation order, so logits, residual, q and KV pages match bit for bit."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    shape = LlamaShape("tiny", 384, 512, 3, 4, 2, 128, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=cb, device=dev, kv_bits=5, max_ctx=2048)
    model.alloc_state(bsz, pos=99)
    model.fold_rotate = False                              # bsz 16: the separate glue_resid + glue_rotate launches are the bit-identical form
    l1 = model.decode_step_fused_v1().clone()
    x1, q1 = model.x.clone(), model.q.clone()
    kv1 = [(c.clone(), s.clone()) for c, s in model.kcache + model.vcache]
    for c, s in model.kcache + model.vcache:
        c.zero_(); s.zero_()
    model.q.zero_(); model.logits.zero_()
    l2 = model.decode_step_fused()
    torch.cuda.synchronize()
    assert torch.equal(l2, l1) and torch.equal(model.x, x1) and torch.equal(model.q, q1)
    for (c, s), (c0, s0) in zip(model.kcache + model.vcache, kv1):
        assert torch.equal(c, c0) and torch.equal(s, s0)


@pytest.mark.parametrize("cb", [0, 2])
@pytest.mark.parametrize("m", [1, 3, 8])
def test_bc_gated_mlp_matches_oracle_and_op_by_op(dev, cb, m):
    """BC_GatedMLP.run_bszN (libtorch/mlp.cpp:14-91 semantics) against the oracle composition and the op-by-op ops."""
    from exllamav3_amd import ext
    hidden, inter, K = 512, 768, 4
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    mats = [o.synth_linear(hidden, inter, K, seed=61, realistic=True), o.synth_linear(hidden, inter, K, seed=62, realistic=True),
            o.synth_linear(inter, hidden, K, seed=63, realistic=True)]
    bcs = [ext.BC_LinearEXL3(T(t[0]), T(t[1]), T(t[2]), K, None, cb == 1, cb == 2, None) for t in mats]
    x = np.random.default_rng(m).standard_normal((1, m, hidden)).astype(np.float16)
    a = torch.empty((1, 8, inter), dtype=torch.half, device=dev); dxh = torch.empty_like(a)
    gu = torch.empty((2, 8, inter), dtype=torch.half, device=dev); guh = torch.empty((2, 8, hidden), dtype=torch.half, device=dev)
    mlp = ext.BC_GatedMLP(guh, gu, a, dxh, None, None, None, K, cb == 1, cb == 2, True, False, False, bcs[0], bcs[1], bcs[2], 0.0)
    for out_dtype in (torch.half, torch.float):
        d = torch.full((1, m, hidden), float("nan"), dtype=out_dtype, device=dev)
        mlp.run_bszN(T(x), d)
        x2 = x.reshape(m, hidden)
        g = o.linear_forward(x2, mats[0][0], mats[0][1], mats[0][2], K, cb).astype(np.float32)
        u = o.linear_forward(x2, mats[1][0], mats[1][1], mats[1][2], K, cb).astype(np.float32)
        av = (g / (1 + np.exp(-g)) * u).astype(np.float16)
        ref = o.linear_forward(av, mats[2][0], mats[2][1], mats[2][2], K, cb, out_fp32=(out_dtype == torch.float)).astype(np.float32)
        got = d.float().cpu().numpy().reshape(m, hidden)
        assert np.isfinite(got).all()
        assert np.abs(got - ref).max() / np.sqrt((ref ** 2).mean()) < 2e-2
        assert np.abs(a[0, :m].float().cpu().numpy() - av.astype(np.float32)).max() < 2e-2 * max(1.0, float(np.abs(av).max()))
        # op-by-op ops of the reference surface
        g2 = torch.empty((m, inter), dtype=torch.half, device=dev); u2 = torch.empty_like(g2); a2 = torch.empty_like(g2)
        bcs[0].run(T(x2), g2); bcs[1].run(T(x2), u2); ext.silu_mul(g2, u2, a2)
        d2 = torch.empty((m, hidden), dtype=out_dtype, device=dev); bcs[2].run(a2, d2)
        assert float((d2.float() - d.float().view(m, hidden)).abs().max()) < 1e-2 * float(d2.float().abs().max()) + 1e-3
    # the other activations of libtorch/mlp.h:64-67 (+ act_limit): launch-per-op route, oracle = the reference's activation kernels restated
    x2 = x.reshape(m, hidden)
    g = o.linear_forward(x2, mats[0][0], mats[0][1], mats[0][2], K, cb); u = o.linear_forward(x2, mats[1][0], mats[1][1], mats[1][2], K, cb)
    for flags, name, limit in (((False, True, False), "gelu", 0.0), ((False, False, True), "relu2", 0.75), ((True, False, False), "silu", 0.5)):
        mlp2 = ext.BC_GatedMLP(guh, gu, a, dxh, None, None, None, K, cb == 1, cb == 2, *flags, bcs[0], bcs[1], bcs[2], limit)
        d = torch.full((1, m, hidden), float("nan"), dtype=torch.half, device=dev)
        mlp2.run_bszN(T(x), d)
        av = o.act_mul(g, u, name, limit)
        ref = o.linear_forward(av, mats[2][0], mats[2][1], mats[2][2], K, cb).astype(np.float32)
        got = d.float().cpu().numpy().reshape(m, hidden)
        assert np.isfinite(got).all() and np.abs(got - ref).max() / np.sqrt((ref ** 2).mean()) < 2e-2, name
    with pytest.raises(RuntimeError):
        ext.BC_GatedMLP(guh, gu, a, dxh, None, None, None, K, False, False, True, True, False, bcs[0], bcs[1], bcs[2], 0.0)      # two activations at once


@pytest.mark.parametrize("hd,hq,hkv", [(128, 4, 2), (64, 8, 2)])
@pytest.mark.parametrize("fused_kv,head_norm", [(False, False), (True, True)])
def test_bc_attention_runner_matches_oracle(dev, hd, hq, hkv, fused_kv, head_norm):
    """BC_Attention.run (libtorch/attention.cpp:246-504, Llama / Mixtral subset): projections -> (head norm +) RoPE -> append to the 4-bit paged
    cache -> decode attention over the quantized cache -> o_proj, against the oracle composition; a pre-filled context of different lengths per
    sequence, positions tensor, the fused k|v pointer-table form, graph replay; unsupported options raise."""
    from exllamav3_amd import ext
    hidden, K, cb, bits, page, bsz, pps = 512, 4, 2, 4, 256, 3, 2
    rng = np.random.default_rng(hd + fused_kv)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    mats = {n: o.synth_linear(k, nn, K, seed=70 + i, realistic=True) for i, (n, k, nn) in enumerate(
        (("q", hidden, hq * hd), ("k", hidden, hkv * hd), ("v", hidden, hkv * hd), ("o", hq * hd, hidden)))}
    dm = {n: tuple(T(a) for a in t) for n, t in mats.items()}
    bc = {n: ext.BC_LinearEXL3(t[0], t[1], t[2], K, None, False, True, None) for n, t in dm.items()}
    G = hkv * hd // 32
    npages = bsz * pps
    bt_np = rng.permutation(npages).reshape(bsz, pps).astype(np.int32)
    lens = np.array([300, 17, 256], np.int32)                                   # tokens already cached (the third ends exactly on a page edge)
    ck = rng.standard_normal((bsz, pps * page, hkv * hd)).astype(np.float16); cv = rng.standard_normal((bsz, pps * page, hkv * hd)).astype(np.float16)
    kq = np.zeros((npages, page, G * bits), np.uint32); ks = np.zeros((npages, page, G), np.float16); vq = kq.copy(); vs = ks.copy()
    for b in range(bsz):
        pk, sc = o.kv_quant(ck[b], bits); pv, sv = o.kv_quant(cv[b], bits)
        for pg in range(pps):
            kq[bt_np[b, pg]] = pk[pg * page:(pg + 1) * page]; ks[bt_np[b, pg]] = sc[pg * page:(pg + 1) * page]
            vq[bt_np[b, pg]] = pv[pg * page:(pg + 1) * page]; vs[bt_np[b, pg]] = sv[pg * page:(pg + 1) * page]
    dkq, dks, dvq, dvs = T(kq.view(np.int32)), T(ks), T(vq.view(np.int32)), T(vs)
    inv_freq = (1.0 / (10000.0 ** (np.arange(0, hd, 2, dtype=np.float32) / hd))).astype(np.float32)
    qn = (1 + 0.1 * rng.standard_normal(hd)).astype(np.float16) if head_norm else None
    kn = (1 + 0.1 * rng.standard_normal(hd)).astype(np.float16) if head_norm else None
    kw = dict(num_q_heads=hq, num_kv_heads=hkv, head_dim=hd, hidden_size=hidden, hidden_size_padded=hidden, page_size=page,
              q_proj=bc["q"], k_proj=bc["k"], v_proj=bc["v"], o_proj=bc["o"], q_norm=T(qn) if head_norm else None, k_norm=T(kn) if head_norm else None,
              norm_eps=1e-6, inv_freq=T(inv_freq), rope_style=2, attn_factor=1.0, quant_cache=True, cache_k=dkq, cache_v=dvq, cache_k_scales=dks,
              cache_v_scales=dvs, xh=None, h32=None)
    if fused_kv:
        ptr = lambda i: torch.tensor([dm["k"][i].data_ptr(), dm["v"][i].data_ptr()], dtype=torch.long, device=dev)
        kw.update(kv_ptrs_trellis=ptr(0), kv_ptrs_suh=ptr(1), kv_ptrs_svh=ptr(2), kv_K=K, kv_mcg=False, kv_mul1=True)
    attn = ext.BC_Attention(**kw)
    assert not attn.needs_configure(bsz, 1)
    x = rng.standard_normal((bsz, 1, hidden)).astype(np.float16)
    positions = (lens + np.array([0, 5, 0])).astype(np.int32)                    # RoPE position need not equal the cache length
    y = torch.full((bsz, 1, hidden), float("nan"), dtype=torch.half, device=dev)
    dl, dbt, dpos = T(lens), T(bt_np), T(positions)
    attn.run(bsz, 1, T(x), y, dl, dbt, 0, dpos, None, None)
    # ---- oracle
    lin = lambda n, a, **k2: o.linear_forward(a, mats[n][0], mats[n][1], mats[n][2], K, cb, **k2)
    x2 = x.reshape(bsz, hidden)
    q, k, v = lin("q", x2), lin("k", x2), lin("v", x2)
    q4, k4 = o.rope(q.reshape(bsz, 1, hq, hd), k.reshape(bsz, 1, hkv, hd), inv_freq, positions=positions, rope_mode=o.ROPE_NEOX,
                    q_norm=qn, k_norm=kn, norm_eps=1e-6)
    kd = np.zeros((bsz, pps * page, hkv, hd), np.float16); vd = np.zeros_like(kd)
    for b in range(bsz):
        full_k = np.concatenate([ck[b, :lens[b]], k4[b].reshape(1, -1)]); full_v = np.concatenate([cv[b, :lens[b]], v[b].reshape(1, -1)])
        pk, sc = o.kv_quant(full_k, bits); pv, sv = o.kv_quant(full_v, bits)
        kd[b, :lens[b] + 1] = o.kv_dequant(pk, sc, bits).reshape(-1, hkv, hd); vd[b, :lens[b] + 1] = o.kv_dequant(pv, sv, bits).reshape(-1, hkv, hd)
    ao = o.attn_decode_qcache(q4.reshape(bsz, hq, hd), kd, vd, lens + 1)
    ref = lin("o", ao.reshape(bsz, -1)).astype(np.float32)
    got = y.float().cpu().numpy().reshape(bsz, hidden)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() / np.sqrt((ref ** 2).mean()) < 3e-2
    # the new token landed in the cache at position lens[b] of every sequence
    row = lambda b: (bt_np[b, lens[b] // page], lens[b] % page)
    for b in range(bsz):
        pk, sc = o.kv_quant(k4[b].reshape(1, -1), bits)
        assert np.allclose(dks.cpu().numpy()[row(b)].astype(np.float32), sc[0].astype(np.float32), rtol=1e-2)   # k itself agrees to GEMV tolerance
    # graph replay of the same call (the appended token is rewritten in place: same bits)
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    dx = T(x)
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            attn.run(bsz, 1, dx, y, dl, dbt, 0, dpos, None, None)
    y.zero_(); g.replay(); torch.cuda.synchronize()
    assert np.array_equal(y.float().cpu().numpy().reshape(bsz, hidden), got)
    with pytest.raises(RuntimeError):
        attn.run(bsz, 17, T(np.zeros((bsz, 17, hidden), np.float16)), y, dl, dbt, 0, dpos, None, None)
    with pytest.raises(RuntimeError):
        ext.BC_Attention(**{**kw, "gate_mode": 2})                              # a full gate without any gate projection (attention.cpp:317)
    with pytest.raises(RuntimeError):
        ext.BC_Attention(**{**kw, "hidden_size_padded": hidden + 64})           # not a multiple of 128
    with pytest.raises(RuntimeError):
        ext.BC_Attention(**{**kw, "quant_cache": False})                        # the cache tensors here are the quantized words, not fp16 pages


@pytest.mark.parametrize("opt", ["v_norm", "k_as_v", "k_as_v_norm", "l4_post_norm", "sinks", "gate_headwise", "gate_headwise_softplus", "gate_full_proj",
                                 "gate_full_weight", "gate_full_fused", "gate_interleaved", "padded_hidden"])
def test_bc_attention_runner_options(dev, opt):
    """The runner's options that compose from this build's ops (attention.cpp:335-395): per-head V norm, V = the K projection's output before head
    norm / RoPE (copied or normed), the rope options (llama-4 query scale + unweighted norm after the rotation), learned attention sinks -- each against the oracle
    composition, over short contexts (a one-token context would hand single 4-bit level flips of the new token's V straight to the output)."""
    from exllamav3_amd import ext
    hidden, K, cb, bits, page, bsz, pps, hd, hq, hkv = 512, 4, 2, 4, 256, 2, 1, 128, 4, 2
    rng = np.random.default_rng(len(opt))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    mats = {n: o.synth_linear(k, nn, K, seed=90 + i, realistic=True) for i, (n, k, nn) in enumerate(
        (("q", hidden, hq * hd), ("k", hidden, hkv * hd), ("v", hidden, hkv * hd), ("o", hq * hd, hidden), ("g", hidden, hq * hd),
         ("qi", hidden, 2 * hq * hd)))}
    dm = {n: tuple(T(a) for a in t) for n, t in mats.items()}
    bc = {n: ext.BC_LinearEXL3(t[0], t[1], t[2], K, None, False, True, None) for n, t in dm.items()}
    gw_head = (rng.standard_normal((hidden, hq)) * 0.08).astype(np.float16)            # fp16 gate weights: one gate per head / per value
    gw_full = (rng.standard_normal((hidden, hq * hd)) * 0.08).astype(np.float16)
    G = hkv * hd // 32
    lens = np.array([40, 23], np.int32)
    bt_np = np.array([[1], [0]], np.int32)
    ck = rng.standard_normal((bsz, page, hkv * hd)).astype(np.float16); cv = rng.standard_normal((bsz, page, hkv * hd)).astype(np.float16)
    kq = np.zeros((bsz, page, G * bits), np.uint32); ks = np.zeros((bsz, page, G), np.float16); vq = kq.copy(); vs = ks.copy()
    for b in range(bsz):
        pk, sc = o.kv_quant(ck[b], bits); pv, sv = o.kv_quant(cv[b], bits)
        kq[bt_np[b, 0]] = pk; ks[bt_np[b, 0]] = sc; vq[bt_np[b, 0]] = pv; vs[bt_np[b, 0]] = sv
    inv_freq = (1.0 / (10000.0 ** (np.arange(0, hd, 2, dtype=np.float32) / hd))).astype(np.float32)
    vw = (1 + 0.1 * rng.standard_normal(hd)).astype(np.float16)
    kw = dict(num_q_heads=hq, num_kv_heads=hkv, head_dim=hd, hidden_size=hidden, hidden_size_padded=hidden, page_size=page,
              q_proj=bc["q"], k_proj=bc["k"], v_proj=bc["v"], o_proj=bc["o"], norm_eps=1e-6, inv_freq=T(inv_freq), rope_style=2, attn_factor=1.0,
              quant_cache=True, cache_k=T(kq.view(np.int32)), cache_v=T(vq.view(np.int32)), cache_k_scales=T(ks), cache_v_scales=T(vs), xh=None, h32=None)
    rope_kw = {}
    if opt == "v_norm":
        kw.update(v_norm=True, v_norm_w=T(vw), v_norm_eps=1e-5, v_norm_constant_bias=0.0, v_norm_constant_scale=1.0)
    elif opt == "k_as_v":
        kw.update(use_k_as_v=True, v_proj=None)
    elif opt == "k_as_v_norm":
        kw.update(use_k_as_v=True, v_proj=None, v_norm=True, v_norm_w=None, v_norm_eps=1e-5)
    elif opt == "sinks":
        sinks = np.linspace(-2.0, 5.0, hq).astype(np.float32)
        kw.update(sinks=T(sinks))
    elif opt.startswith("gate_headwise"):
        kw.update(gate_mode=1, gate_softplus=opt.endswith("softplus"), g_weight=T(gw_head))
    elif opt == "gate_full_proj":
        kw.update(gate_mode=2, g_proj=bc["g"])
    elif opt == "gate_full_weight":
        kw.update(gate_mode=2, g_weight=T(gw_full))
    elif opt == "gate_full_fused":
        ptr = lambda i: torch.tensor([dm["q"][i].data_ptr(), dm["g"][i].data_ptr()], dtype=torch.long, device=dev)
        kw.update(gate_mode=2, qg_ptrs_trellis=ptr(0), qg_ptrs_suh=ptr(1), qg_ptrs_svh=ptr(2), qg_K=K, qg_mcg=False, qg_mul1=True)
    elif opt == "gate_interleaved":
        kw.update(gate_mode=3, q_proj=bc["qi"])
    elif opt == "padded_hidden":
        kw.update(hidden_size=hidden - 32)                                  # the model's width is 480, the quantized linears are padded to 512
    else:
        kw.update(l4_scaling_beta=0.3, l4_scaling_original=16, post_rope_norm=True)
        rope_kw = dict(l4_beta=0.3, l4_orig=16, post_rope_norm=True)
    attn = ext.BC_Attention(**kw)
    x = rng.standard_normal((bsz, 1, hidden)).astype(np.float16)
    positions = np.array([40, 30], np.int32)
    hw = hidden - 32 if opt == "padded_hidden" else hidden                 # width of x and y as the caller sees them
    if opt == "padded_hidden":
        x[..., hw:] = 0
    y = torch.full((bsz, 1, hw), float("nan"), dtype=torch.half, device=dev)
    attn.run(bsz, 1, T(x[..., :hw]), y, T(lens), T(bt_np), 0, T(positions), None, None)
    lin = lambda n, a: o.linear_forward(a, mats[n][0], mats[n][1], mats[n][2], K, cb)
    x2 = x.reshape(bsz, hidden)
    q, k = lin("q", x2), lin("k", x2)
    gate = None
    if opt == "gate_interleaved":
        q, gate = o.deinterleave_qg(lin("qi", x2), hd)
    elif opt in ("gate_full_proj", "gate_full_fused"):
        gate = lin("g", x2)
    elif opt == "gate_full_weight":
        gate = (x2.astype(np.float32) @ gw_full.astype(np.float32)).astype(np.float16)
    elif opt.startswith("gate_headwise"):
        gate = (x2.astype(np.float32) @ gw_head.astype(np.float32)).astype(np.float16)
    v = k.copy() if opt.startswith("k_as_v") else lin("v", x2)
    if opt in ("v_norm", "k_as_v_norm"):
        v = o.rms_norm(v.reshape(bsz * hkv, hd), vw if opt == "v_norm" else None, 1e-5).reshape(bsz, hkv * hd)
    q4, k4 = o.rope(q.reshape(bsz, 1, hq, hd), k.reshape(bsz, 1, hkv, hd), inv_freq, positions=positions, rope_mode=o.ROPE_NEOX, norm_eps=1e-6, **rope_kw)
    kd = np.zeros((bsz, page, hkv, hd), np.float16); vd = np.zeros_like(kd)
    for b in range(bsz):
        full_k = np.concatenate([ck[b, :lens[b]], k4[b].reshape(1, -1)]); full_v = np.concatenate([cv[b, :lens[b]], v[b].reshape(1, -1)])
        pk, sc = o.kv_quant(full_k, bits); pv, sv = o.kv_quant(full_v, bits)
        kd[b, :lens[b] + 1] = o.kv_dequant(pk, sc, bits).reshape(-1, hkv, hd); vd[b, :lens[b] + 1] = o.kv_dequant(pv, sv, bits).reshape(-1, hkv, hd)
    ao = o.attn_decode_qcache(q4.reshape(bsz, hq, hd), kd, vd, lens + 1, sinks=sinks if opt == "sinks" else None)
    if opt == "gate_headwise_softplus":
        ao = o.mul_softplus_broadcast(ao, gate)                            # (bsz, hq, hd) x (bsz, hq)
    elif opt == "gate_headwise":
        ao = o.mul_sigmoid(ao, gate)
    elif gate is not None:
        ao = o.mul_sigmoid(ao.reshape(bsz, -1), gate)
    ref = lin("o", ao.reshape(bsz, -1)).astype(np.float32)[:, :hw]
    got = y.float().cpu().numpy().reshape(bsz, hw)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() / np.sqrt((ref ** 2).mean()) < 3e-2


@pytest.mark.parametrize("hd,hq,hkv,q_len", [(128, 4, 2, 1), (64, 8, 2, 1), (128, 8, 2, 7)])
def test_bc_attention_runner_fp16_cache(dev, hd, hq, hkv, q_len):
    """BC_Attention over an UNQUANTIZED paged cache (quant_cache = False, attention.cpp:400-418): the new rows land in the fp16 pages at
    cache_seqlens[b] + t, attention runs causally over the pages; one new token and a 7-token chunk, contexts ending inside / on a page edge and an
    empty one, scattered pages; against the oracle (fp32 softmax attention over the same fp16 rows); graph replay reproduces the bits."""
    from exllamav3_amd import ext
    hidden, K, cb, page, bsz, pps = 512, 4, 2, 256, 3, 2
    rng = np.random.default_rng(hd + q_len)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    mats = {n: o.synth_linear(k, nn, K, seed=110 + i, realistic=True) for i, (n, k, nn) in enumerate(
        (("q", hidden, hq * hd), ("k", hidden, hkv * hd), ("v", hidden, hkv * hd), ("o", hq * hd, hidden)))}
    bc = {n: ext.BC_LinearEXL3(T(t[0]), T(t[1]), T(t[2]), K, None, False, True, None) for n, t in mats.items()}
    npages = bsz * pps
    bt_np = rng.permutation(npages).reshape(bsz, pps).astype(np.int32)
    lens = np.array([300, 0, 256 - q_len if q_len > 1 else 256], np.int32)
    ck = rng.standard_normal((bsz, pps * page, hkv, hd)).astype(np.float16); cv = rng.standard_normal((bsz, pps * page, hkv, hd)).astype(np.float16)
    kp = np.zeros((npages, page, hkv, hd), np.float16); vp = np.zeros_like(kp)
    for b in range(bsz):
        for pg in range(pps):
            kp[bt_np[b, pg]] = ck[b, pg * page:(pg + 1) * page]; vp[bt_np[b, pg]] = cv[b, pg * page:(pg + 1) * page]
    inv_freq = (1.0 / (10000.0 ** (np.arange(0, hd, 2, dtype=np.float32) / hd))).astype(np.float32)
    dkp, dvp = T(kp), T(vp)
    attn = ext.BC_Attention(num_q_heads=hq, num_kv_heads=hkv, head_dim=hd, hidden_size=hidden, hidden_size_padded=hidden, page_size=page,
                            q_proj=bc["q"], k_proj=bc["k"], v_proj=bc["v"], o_proj=bc["o"], norm_eps=1e-6, inv_freq=T(inv_freq), rope_style=2, attn_factor=1.0,
                            quant_cache=False, cache_k=dkp, cache_v=dvp, xh=None, h32=None)
    x = rng.standard_normal((bsz, q_len, hidden)).astype(np.float16)
    y = torch.full((bsz, q_len, hidden), float("nan"), dtype=torch.half, device=dev)
    dl, dbt = T(lens), T(bt_np)
    attn.run(bsz, q_len, T(x), y, dl, dbt, 0, dl, None, None)                 # rope positions = positions in the cache
    lin = lambda n, a: o.linear_forward(a, mats[n][0], mats[n][1], mats[n][2], K, cb)
    x2 = x.reshape(bsz * q_len, hidden)
    q, k, v = lin("q", x2), lin("k", x2), lin("v", x2)
    q4, k4 = o.rope(q.reshape(bsz, q_len, hq, hd), k.reshape(bsz, q_len, hkv, hd), inv_freq, positions=lens, rope_mode=o.ROPE_NEOX, norm_eps=1e-6)
    v4 = v.reshape(bsz, q_len, hkv, hd)
    ao = np.zeros((bsz, q_len, hq, hd), np.float16)
    for b in range(bsz):
        fk = np.concatenate([ck[b, :lens[b]], k4[b]]); fv = np.concatenate([cv[b, :lens[b]], v4[b]])
        for t in range(q_len):
            n = lens[b] + t + 1
            ao[b, t] = o.attn_decode_qcache(q4[b, t][None], fk[None, :n], fv[None, :n], [n])[0]
    ref = lin("o", ao.reshape(bsz * q_len, -1)).astype(np.float32)
    got = y.float().cpu().numpy().reshape(bsz * q_len, hidden)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() / np.sqrt((ref ** 2).mean()) < 2e-2
    # the appended rows are where the block table says (K after rope, to GEMV tolerance)
    for b in range(bsz):
        for t in (0, q_len - 1):
            pos = lens[b] + t
            row = dkp[bt_np[b, pos // page], pos % page].float().cpu().numpy()
            assert np.abs(row - k4[b, t].astype(np.float32)).max() < 3e-2 * max(1.0, np.abs(k4[b, t].astype(np.float32)).max())
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    dx = T(x)
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            attn.run(bsz, q_len, dx, y, dl, dbt, 0, dl, None, None)
    y.zero_(); g.replay(); torch.cuda.synchronize()
    assert np.array_equal(y.float().cpu().numpy().reshape(bsz * q_len, hidden), got)
    with pytest.raises(RuntimeError):
        ext.BC_Attention(num_q_heads=hq, num_kv_heads=hkv, head_dim=hd, hidden_size=hidden, hidden_size_padded=hidden, page_size=page, q_proj=bc["q"], k_proj=bc["k"],
                         v_proj=bc["v"], o_proj=bc["o"], inv_freq=T(inv_freq), quant_cache=False, cache_k=dkp, cache_v=dvp, sinks=torch.zeros(hq, device=dev))


@pytest.mark.parametrize("hd,hq,hkv,q_len,fused_kv", [(128, 4, 2, 5, False), (64, 8, 2, 16, True), (128, 8, 2, 16, False)])
def test_bc_attention_runner_multi_token(dev, hd, hq, hkv, q_len, fused_kv):
    """BC_Attention.run with 2 <= q_len <= 16 (attention.cpp:246-504; the runner's MAX_QLEN): the chunk is appended to the 4-bit paged cache, the
    block table's pages are expanded to fp16 and the chunk attends causally over context + itself; against the oracle composition
    (projections, rope at positions[b] + t, quantize -> dequantize round trip of the whole sequence, causal attention, o_proj)."""
    from exllamav3_amd import ext
    hidden, K, cb, bits, page, bsz, pps = 512, 4, 2, 4, 256, 3, 3
    rng = np.random.default_rng(hd + q_len)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    mats = {n: o.synth_linear(k, nn, K, seed=170 + i, realistic=True) for i, (n, k, nn) in enumerate(
        (("q", hidden, hq * hd), ("k", hidden, hkv * hd), ("v", hidden, hkv * hd), ("o", hq * hd, hidden)))}
    dm = {n: tuple(T(a) for a in t) for n, t in mats.items()}
    bc = {n: ext.BC_LinearEXL3(t[0], t[1], t[2], K, None, False, True, None) for n, t in dm.items()}
    G = hkv * hd // 32
    npages = bsz * pps
    bt_np = rng.permutation(npages).reshape(bsz, pps).astype(np.int32)
    lens = np.array([300, 0, 256 - 3], np.int32)                                # empty context; a chunk that crosses a page edge
    ck = rng.standard_normal((bsz, pps * page, hkv * hd)).astype(np.float16); cv = rng.standard_normal((bsz, pps * page, hkv * hd)).astype(np.float16)
    kq = np.zeros((npages, page, G * bits), np.uint32); ks = np.zeros((npages, page, G), np.float16); vq = kq.copy(); vs = ks.copy()
    for b in range(bsz):
        pk, sc = o.kv_quant(ck[b], bits); pv, sv = o.kv_quant(cv[b], bits)
        for pg in range(pps):
            kq[bt_np[b, pg]] = pk[pg * page:(pg + 1) * page]; ks[bt_np[b, pg]] = sc[pg * page:(pg + 1) * page]
            vq[bt_np[b, pg]] = pv[pg * page:(pg + 1) * page]; vs[bt_np[b, pg]] = sv[pg * page:(pg + 1) * page]
    dkq, dks, dvq, dvs = T(kq.view(np.int32)), T(ks), T(vq.view(np.int32)), T(vs)
    inv_freq = (1.0 / (10000.0 ** (np.arange(0, hd, 2, dtype=np.float32) / hd))).astype(np.float32)
    kw = dict(num_q_heads=hq, num_kv_heads=hkv, head_dim=hd, hidden_size=hidden, hidden_size_padded=hidden, page_size=page,
              q_proj=bc["q"], k_proj=bc["k"], v_proj=bc["v"], o_proj=bc["o"], norm_eps=1e-6, inv_freq=T(inv_freq), rope_style=2, attn_factor=1.0,
              quant_cache=True, cache_k=dkq, cache_v=dvq, cache_k_scales=dks, cache_v_scales=dvs, xh=None, h32=None)
    if fused_kv:
        ptr = lambda i: torch.tensor([dm["k"][i].data_ptr(), dm["v"][i].data_ptr()], dtype=torch.long, device=dev)
        kw.update(kv_ptrs_trellis=ptr(0), kv_ptrs_suh=ptr(1), kv_ptrs_svh=ptr(2), kv_K=K, kv_mcg=False, kv_mul1=True)
    attn = ext.BC_Attention(**kw)
    x = rng.standard_normal((bsz, q_len, hidden)).astype(np.float16)
    y = torch.full((bsz, q_len, hidden), float("nan"), dtype=torch.half, device=dev)
    dl, dbt, dpos = T(lens), T(bt_np), T(lens)
    attn.run(bsz, q_len, T(x), y, dl, dbt, 0, dpos, None, None)
    lin = lambda n, a, **k2: o.linear_forward(a, mats[n][0], mats[n][1], mats[n][2], K, cb, **k2)
    x2 = x.reshape(bsz * q_len, hidden)
    q, k, v = lin("q", x2), lin("k", x2), lin("v", x2)
    pos_ids = (lens[:, None] + np.arange(q_len)[None, :]).astype(np.int32)
    q4, k4 = o.rope(q.reshape(bsz, q_len, hq, hd), k.reshape(bsz, q_len, hkv, hd), inv_freq, position_ids=pos_ids, rope_mode=o.ROPE_NEOX)
    v4 = v.reshape(bsz, q_len, hkv * hd)
    ao = np.zeros((bsz, q_len, hq, hd), np.float32)
    for b in range(bsz):
        full_k = np.concatenate([ck[b, :lens[b]], k4[b].reshape(q_len, -1)]); full_v = np.concatenate([cv[b, :lens[b]], v4[b]])
        pk, sc = o.kv_quant(full_k, bits); pv, sv = o.kv_quant(full_v, bits)
        kd = o.kv_dequant(pk, sc, bits).reshape(1, -1, hkv, hd); vd = o.kv_dequant(pv, sv, bits).reshape(1, -1, hkv, hd)
        ao[b] = o.attn_prefill(q4[b:b + 1], kd, vd, np.array([lens[b] + q_len]))[0]
    ref = lin("o", ao.astype(np.float16).reshape(bsz * q_len, -1)).astype(np.float32)
    got = y.float().cpu().numpy().reshape(bsz * q_len, hidden)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() / np.sqrt((ref ** 2).mean()) < 3e-2
    # graph replay (the chunk is rewritten in place with the same bits)
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    dx = T(x)
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            attn.run(bsz, q_len, dx, y, dl, dbt, 0, dpos, None, None)
    y.zero_(); g.replay(); torch.cuda.synchronize()
    assert np.array_equal(y.float().cpu().numpy().reshape(bsz * q_len, hidden), got)


def test_prefill_reconstruct_ahead_is_bit_identical_to_inline(dev):
    """ReconstructAhead (W of the next Linears rebuilt on a side stream, ring of 3 buffers, event-ordered) must not change a bit:
    3 layers x 7 linears cycle the ring several times; repeated chunks reuse the scheduler object."""
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    shape = LlamaShape("tiny", 256, 512, 3, 4, 2, 128, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4)
    model.reconstruct_ahead = False
    ref = model.prefill_chunk(1024).clone()
    kv_ref = [c.clone() for c in model.pf_cache]
    model.reconstruct_ahead = True
    for rep in range(3):
        for c in model.pf_cache: c.zero_()
        got = model.prefill_chunk(1024)
        torch.cuda.synchronize()
        assert torch.equal(got, ref), f"rep {rep}"
        for a, b in zip(model.pf_cache, kv_ref):
            assert torch.equal(a, b)


@pytest.mark.parametrize("bsz", [1, 3, 9])
def test_fused_pipeline_head_dim_64(dev, bsz):
    """Llama-3.2-1B geometry (head_dim 64: two heads per Hadamard block) through the fused pipeline: glue_qkv's RoPE pairs (d, d + 32) and
    the KV groups must land exactly where the op-by-op pipeline puts them; logits against the oracle."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    shape = LlamaShape("tiny64", 256, 512, 2, 8, 4, 64, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=2048)
    model.alloc_state(bsz, pos=777)
    lu = model.decode_step().float().cpu().numpy().copy()
    qu = model.q.clone()
    ku = [(c.clone(), s.clone()) for c, s in model.kcache + model.vcache]
    for c, s in model.kcache + model.vcache:
        c.zero_(); s.zero_()
    lf = model.decode_step_fused().float().cpu().numpy()
    rms = np.sqrt((lu ** 2).mean())
    assert np.abs(lf - lu).max() / rms < 1e-2
    for (c, s), (c0, s0) in zip(model.kcache + model.vcache, ku):
        assert bool(((s != 0) == (s0 != 0)).all())                        # same slots written
        d1 = torch.empty((c.shape[0] * c.shape[1], c.shape[2] // 4 * 32), dtype=torch.half, device=dev); d0 = torch.empty_like(d1)
        ext.dequant_cache_cont(c.view(d1.shape[0], -1), s.view(d1.shape[0], -1), d1)
        ext.dequant_cache_cont(c0.view(d1.shape[0], -1), s0.view(d1.shape[0], -1), d0)
        assert float((d1.float() - d0.float()).abs().max()) < 0.35
    if bsz <= 3:
        ref = _oracle_decode(model, _np(model.x0))
        assert np.abs(lf - ref).max() / np.sqrt((ref ** 2).mean()) < 3e-2


@pytest.mark.parametrize("m", [1, 5, 8])
def test_in_gemv_rmsnorm_hidden_8192(dev, m):
    """Llama-70B width (64 Hadamard blocks per row: more than one 32-lane pass over the per-block sums of squares): glue_resid +
    GEMV_IN_NORM and glue_resid + glue_rotate + rotated GEMV against glue_norm + rotated GEMV, bit for bit."""
    from exllamav3_amd import ext
    ext.set_gemm3_min_rows(0)          # bit-for-bit needs one kernel generation on all three routes (GEMV_IN_NORM exists in generation 2 only)
    k, n, K = 8192, 256, 3
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rng = np.random.default_rng(m)
    tr, suh, svh = o.synth_linear(k, n, K, realistic=True)
    ttr, tsu, tsv = T(tr), T(suh), T(svh)
    w = T((1 + 0.1 * rng.standard_normal(k)).astype(np.float16))
    r0 = T((rng.standard_normal((m, k)) * 2.5).astype(np.float16))
    xh = torch.empty((m, k), dtype=torch.half, device=dev); xs = torch.empty((m, k // 128), dtype=torch.float32, device=dev)
    ext.glue_norm(None, 0, None, None, r0, w, 1e-5, [tsu], [xh], [xs], m)
    y0 = torch.empty((m, n), dtype=torch.half, device=dev)
    ext.exl3_gemv_ex(None, [xh], [xs], [ttr], [y0], None, [tsv], m, False, False, ext.GEMV_IN_ROTATED)
    ss = torch.empty((m, k // 128), dtype=torch.float32, device=dev)
    ext.glue_resid(None, 0, None, None, r0, ss, m)
    y1 = torch.empty_like(y0)
    ext.exl3_gemv_ex_norm(r0, w, ss, 1e-5, [ttr], [y1], [tsu], [tsv], m, False, False, 0)
    xh2 = torch.empty_like(xh)
    ext.glue_rotate(r0, ss, w, 1e-5, [tsu], [xh2], m)
    y2 = torch.empty_like(y0)
    ext.exl3_gemv_ex(None, [xh2], None, [ttr], [y2], None, [tsv], m, False, False, ext.GEMV_IN_ROTATED)
    assert torch.equal(xh, xh2)
    assert torch.equal(y0, y1) and torch.equal(y0, y2)
    ref = o.linear_forward(o.rms_norm(r0.cpu().numpy(), w.cpu().numpy(), 1e-5), tr, suh, svh, K, 0).astype(np.float32)
    assert np.abs(y1.float().cpu().numpy() - ref).max() / np.sqrt((ref ** 2).mean()) < 1e-2


@pytest.mark.parametrize("cb", [0, 2])
@pytest.mark.parametrize("tokens", [1, 3])
def test_moe_block_matches_oracle(dev, cb, tokens):
    """Router (routing_std) + indexed / weighted exl3_mgemm composition (moe_path.SyntheticEXL3MoE) against the oracle: same experts
    selected, weights within fp16, output within 2e-2 RMS."""
    from exllamav3_amd.moe_path import SyntheticEXL3MoE
    moe = SyntheticEXL3MoE(256, 384, experts=6, top_k=2, K=4, cb=cb, device=dev, seed=5)
    x = torch.randn((tokens, 256), device=dev, generator=torch.Generator(device=dev).manual_seed(tokens)).half()
    y = moe.forward(x).float().cpu().numpy()
    xs = x.cpu().numpy()
    scores, sel, w = o.routing_std(xs, moe.router.cpu().numpy(), 2)
    assert np.abs(moe.scores.float().cpu().numpy() - scores.astype(np.float32)).max() < 2e-3 * max(1.0, float(np.abs(scores).max()))
    assert np.array_equal(moe.sel.cpu().numpy(), sel)
    assert np.abs(moe.w.float().cpu().numpy() - w.astype(np.float32)).max() < 2e-3
    ref = np.zeros((tokens, 256), dtype=np.float32)
    for t in range(tokens):
        for j in range(2):
            e = int(sel[t, j])
            g = _lin(moe.gate[e], xs[t:t + 1]).astype(np.float32); u = _lin(moe.up[e], xs[t:t + 1]).astype(np.float32)
            a = (g / (1 + np.exp(-g)) * u).astype(np.float16)
            ref[t] += float(w[t, j]) * _lin(moe.down[e], a, out_fp32=True)[0]
    assert np.abs(y - ref).max() / np.sqrt((ref ** 2).mean()) < 2e-2


def test_exl3_moe_op_matches_oracle(dev):
    """ext.exl3_moe (quant/exl3_moe.cu:99-301, same argument list): assignments grouped by expert, ragged counts -- an expert with more than 16
    rows (two slots of the indexed launches), single-row experts, idle experts, and one above max_tokens_per_expert that the op must leave
    alone -- accumulated into a pre-filled fp32 output; against the oracle's per-assignment composition."""
    from exllamav3_amd import ext
    from exllamav3_amd.moe_path import SyntheticEXL3MoE
    hidden, inter, E, bsz, max_rows = 256, 384, 6, 24, 22
    moe = SyntheticEXL3MoE(hidden, inter, experts=E, top_k=2, K=4, cb=2, device=dev, seed=9)
    rng = np.random.default_rng(4)
    counts = [20, 1, 0, 23, 4, 0]                                            # expert 3 exceeds max_rows: skipped by the op
    toks, wts = [], []
    for c in counts:
        toks += list(rng.permutation(bsz)[:c]); wts += list(rng.uniform(0.1, 0.9, c))
    x = rng.standard_normal((bsz, hidden)).astype(np.float16)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out0 = rng.standard_normal((bsz, hidden)).astype(np.float32)              # the op accumulates
    out = T(out0)
    tmp_s = torch.empty((4, max_rows, hidden), dtype=torch.half, device=dev); tmp_i = torch.empty((4, max_rows, inter), dtype=torch.half, device=dev)
    wts16 = np.array(wts, np.float16)
    args = (T(x), out, T(np.array(counts + [0], np.int64)), T(np.array(toks, np.int64)), T(wts16), tmp_s, tmp_s, tmp_i, tmp_i, ext.MOE_ACT_SILU,
            4, 4, 4, moe.g_B, moe.g_suh, moe.g_svh, moe.u_B, moe.u_suh, moe.u_svh, moe.d_B, moe.d_suh, moe.d_svh, False, True, False, True, False, True)
    ext.exl3_moe(*args, 0.0, 3)
    ref = out0.copy()
    off = 0
    for e, c in enumerate(counts):
        if 0 < c <= max_rows:
            rows = np.array(toks[off:off + c])
            g = _lin(moe.gate[e], x[rows]).astype(np.float32); u = _lin(moe.up[e], x[rows]).astype(np.float32)
            a = (g / (1 + np.exp(-g)) * u).astype(np.float16)
            d = _lin(moe.down[e], a, out_fp32=True)
            for i, t in enumerate(rows):
                ref[t] += np.float32(wts16[off + i]) * d[i]
        off += c
    got = out.cpu().numpy()
    delta = ref - out0
    assert np.abs(got - ref).max() / np.sqrt((delta[np.abs(delta).sum(1) > 0] ** 2).mean()) < 2e-2
    untouched = np.setdiff1d(np.arange(bsz), np.array([t for e, c in enumerate(counts) if 0 < c <= max_rows
                                                         for t in toks[sum(counts[:e]): sum(counts[:e]) + c]]))
    assert np.array_equal(got[untouched], out0[untouched])
    ext.exl3_moe(*args, 0.0, 0)                                              # num_active == 0: nothing to do
    assert np.array_equal(out.cpu().numpy(), got)
    # the other activations of the reference's fused op (exl3_moe_common.cuh:6-8, hadamard_inner.cuh:343-397): GELU, the non-gated relu^2 (no gate
    # GEMM: relu(u) * u), an activation limit (u clamped to +-limit, the activated gate to <= limit)
    for act, name, limit in ((ext.MOE_ACT_GELU, "gelu", 0.0), (ext.MOE_ACT_RELU2_NOGATE, "relu", 0.0), (ext.MOE_ACT_SILU, "silu", 0.75),
                             (ext.MOE_ACT_RELU2_NOGATE, "relu", 0.5)):
        out = T(out0)
        ext.exl3_moe(*args[:1], out, *args[2:9], act, *args[10:], limit, 3)
        ref = out0.copy()
        off = 0
        for e, c in enumerate(counts):
            if 0 < c <= max_rows:
                rows = np.array(toks[off:off + c])
                u = _lin(moe.up[e], x[rows]).astype(np.float16)
                g = _lin(moe.gate[e], x[rows]).astype(np.float16) if act != ext.MOE_ACT_RELU2_NOGATE else u
                a = o.act_mul(g, u, name, limit)
                d = _lin(moe.down[e], a, out_fp32=True)
                for i, t in enumerate(rows):
                    ref[t] += np.float32(wts16[off + i]) * d[i]
            off += c
        got2 = out.cpu().numpy()
        delta = ref - out0
        assert np.abs(got2 - ref).max() / np.sqrt((delta[np.abs(delta).sum(1) > 0] ** 2).mean()) < 2e-2, (name, limit)
        assert np.array_equal(got2[untouched], out0[untouched])
    with pytest.raises(RuntimeError):
        ext.exl3_moe(*args[:9], 7, *args[10:], 0.0, 3)
    assert ext.exl3_moe_max_concurrency(0) > 0


@pytest.mark.parametrize("act", ["silu", "gelu", "relu2_nogate"])
def test_bc_blocksparse_mlp_runner_matches_oracle(dev, act):
    """ext.BC_BlockSparseMLP (libtorch/blocksparse_mlp.h:24-205; constructed as modules/block_sparse_mlp.py:800-975 constructs it, the same statics):
    run_bszN for 1 and 3 tokens (indexed gate / up launches over the selected experts, activation, weighted indexed down launch: rows 0 .. tokens - 1 of
    out_d), the same under an expert range (the partial sum over the rank's experts), run_single_expert (5 rows through the quantized kernels),
    run_single_expert_dq (40 rows through reconstruct + hgemm) -- each against the oracle's per-expert composition."""
    from exllamav3_amd import ext
    from exllamav3_amd.moe_path import SyntheticEXL3MoE
    H, I, E, top, K = 256, 384, 6, 2, 4
    moe = SyntheticEXL3MoE(H, I, experts=E, top_k=top, K=K, cb=2, device=dev, seed=21)
    gated = act != "relu2_nogate"
    rng = np.random.default_rng(len(act))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    MAXB, ROWS = 16, 32
    bszn = MAXB * top
    temp_hidden = torch.empty((max(ROWS * 2, bszn), H), dtype=torch.half, device=dev)
    temp_interm = torch.empty((max(ROWS * 2, 2 * bszn), I), dtype=torch.half, device=dev)
    temp_activa = torch.empty((max(ROWS, bszn), I), dtype=torch.half, device=dev)
    temp_output = torch.empty((max(ROWS, bszn), H), dtype=torch.float, device=dev)
    bcl = lambda l: ext.BC_LinearEXL3(l.trellis, l.suh, l.svh, K, None, l.mcg, l.mul1, None)
    gates, ups, downs = [bcl(l) for l in moe.gate], [bcl(l) for l in moe.up], [bcl(l) for l in moe.down]

    def make(min_e=-1, max_e=-1):
        # an expert range means the rank HOLDS only those experts: its pointer tables and handle lists are local, the kernels rebase the routed
        # index by min_expert (exl3_gemm_kernel.cuh:99-127; modules/block_sparse_mlp.py builds MultiLinear over the local experts)
        lo, hi = (0, E) if min_e < 0 else (min_e, max_e)
        loc = lambda t: t[lo:hi].contiguous()
        return _make(min_e, max_e, lo, hi, loc)

    def _make(min_e, max_e, lo, hi, loc):
        return ext.BC_BlockSparseMLP(
            temp_hidden, temp_hidden[:bszn].view(bszn, 1, H), temp_interm, temp_interm[:bszn].view(bszn, 1, I), temp_interm[bszn:2 * bszn].view(bszn, 1, I),
            temp_activa[:bszn].view(bszn, 1, I), temp_activa, temp_output[:bszn].view(bszn, 1, H), temp_output, None, None,
            torch.empty((H, I), dtype=torch.half, device=dev), torch.empty((I, H), dtype=torch.half, device=dev), min_e, max_e,
            loc(moe.g_B if gated else moe.u_B), loc(moe.g_suh if gated else moe.u_suh), loc(moe.g_svh if gated else moe.u_svh), K, False, True,
            loc(moe.u_B), loc(moe.u_suh), loc(moe.u_svh), K, False, True, loc(moe.d_B), loc(moe.d_suh), loc(moe.d_svh), K, False, True,
            act == "silu", act == "gelu", False, None, None, 0.0, gates[lo:hi] if gated else [], ups[lo:hi], downs[lo:hi],
            None, None, None, torch.empty((bszn, H), dtype=torch.half, device=dev), act_relu2=not gated)

    def expert(e, xr):
        u = _lin(moe.up[e], xr).astype(np.float16)
        g = _lin(moe.gate[e], xr).astype(np.float16) if gated else u
        a = o.act_mul(g, u, {"silu": "silu", "gelu": "gelu", "relu2_nogate": "relu"}[act], 0.0)
        return _lin(moe.down[e], a, out_fp32=True)

    bc = make()
    for bsz in (1, 3):
        x = rng.standard_normal((bsz, H)).astype(np.float16)
        sel = np.stack([rng.permutation(E)[:top] for _ in range(bsz)]).astype(np.int64)
        wts = rng.uniform(0.2, 0.8, (bsz, top)).astype(np.float16)
        bc.run_bszN(T(x), T(sel), T(wts))
        got = temp_output[:bsz].cpu().numpy()
        ref = np.zeros((bsz, H), np.float32)
        for t in range(bsz):
            for j in range(top):
                ref[t] += np.float32(wts[t, j]) * expert(int(sel[t, j]), x[t:t + 1])[0]
        assert np.abs(got - ref).max() / np.sqrt((ref ** 2).mean()) < 2e-2, bsz
    # expert parallelism: this "rank" owns experts [2, 5) -- the partial sum over them (exl3_gemm_kernel.cuh:99-127); one token
    bc_ep = make(2, 5)
    x = rng.standard_normal((1, H)).astype(np.float16)
    sel = np.array([[3, 0]], np.int64); wts = np.array([[0.6, 0.4]], np.float16)
    bc_ep.run_bszN(T(x), T(sel), T(wts))
    ref = np.float32(wts[0, 0]) * expert(3, x)[0]
    got = temp_output[:1].cpu().numpy()[0]
    assert np.abs(got - ref).max() / np.sqrt((ref ** 2).mean()) < 2e-2
    # one expert, a few rows, quantized kernels
    xr = rng.standard_normal((5, H)).astype(np.float16)
    bc.run_single_expert(T(xr), 1)
    ref = expert(1, xr)
    assert np.abs(temp_output[:5].cpu().numpy() - ref).max() / np.sqrt((ref ** 2).mean()) < 2e-2
    # one expert, many rows, reconstruct + hgemm
    R = 40
    xr = rng.standard_normal((R, H)).astype(np.float16)
    yh = torch.empty((2 * R, H), dtype=torch.half, device=dev); interm = torch.empty((2 * R, I), dtype=torch.half, device=dev)
    ia = torch.empty((R, I), dtype=torch.half, device=dev); out = torch.empty((R, H), dtype=torch.half, device=dev)
    bc.run_single_expert_dq(T(xr), 4, yh, interm, ia, out)
    ref = expert(4, xr)
    assert np.abs(out.float().cpu().numpy() - ref).max() / np.sqrt((ref ** 2).mean()) < 3e-2
    with pytest.raises(RuntimeError):
        bc.run_bszN(T(np.zeros((17, H), np.float16)), T(np.zeros((17, top), np.int64)), T(np.zeros((17, top), np.float16)))


@pytest.mark.parametrize("gate", [False, True])
def test_bc_blocksparse_mlp_shared_expert_and_bc_mlp(dev, gate):
    """BC_BlockSparseMLP with a shared expert (a BC_GatedMLP run on every token, blocksparse_mlp.cpp:228-246): plain add, or merged through the
    sigmoid of the token's projection onto an fp16 gate vector (add_sigmoid_gate_proj); and BC_MLP (libtorch/mlp.h:116-190: the non-gated MLP's
    bsz-1 runner) with a padded input width -- against the oracle composition."""
    from exllamav3_amd import ext
    from exllamav3_amd.moe_path import SyntheticEXL3MoE
    H, I, E, top, K = 256, 384, 4, 2, 4
    moe = SyntheticEXL3MoE(H, I, experts=E, top_k=top, K=K, cb=2, device=dev, seed=33)
    rng = np.random.default_rng(int(gate))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    MAXB = ext.BC_BlockSparseMLP.MAX_BSZN
    bszn = MAXB * top
    th = torch.empty((64, H), dtype=torch.half, device=dev); ti = torch.empty((64, I), dtype=torch.half, device=dev)
    ta = torch.empty((32, I), dtype=torch.half, device=dev); to_ = torch.empty((32, H), dtype=torch.float, device=dev)
    bcl = lambda l: ext.BC_LinearEXL3(l.trellis, l.suh, l.svh, K, None, l.mcg, l.mul1, None)
    # the shared expert: one more gated MLP of the same shape
    sh = SyntheticEXL3MoE(H, I, experts=1, top_k=1, K=K, cb=2, device=dev, seed=34)
    shg, shu, shd = bcl(sh.gate[0]), bcl(sh.up[0]), bcl(sh.down[0])
    sh_bc = ext.BC_GatedMLP(torch.empty((2, MAXB, H), dtype=torch.half, device=dev), torch.empty((2, MAXB, I), dtype=torch.half, device=dev),
                            torch.empty((1, MAXB, I), dtype=torch.half, device=dev), torch.empty((1, MAXB, I), dtype=torch.half, device=dev),
                            None, None, None, K, False, True, True, False, False, shg, shu, shd, 0.0)
    gw = (rng.standard_normal((H, 1)) * 0.1).astype(np.float16)
    sh_gate = ext.BC_LinearFP16(T(gw)) if gate else None
    out_d_sh = torch.empty((1, MAXB, H), dtype=torch.float, device=dev)
    bc = ext.BC_BlockSparseMLP(
        th, th[:bszn].view(bszn, 1, H), ti, ti[:bszn].view(bszn, 1, I), ti[bszn:2 * bszn].view(bszn, 1, I), ta[:bszn].view(bszn, 1, I), ta,
        to_[:bszn].view(bszn, 1, H), to_, out_d_sh, None, torch.empty((H, I), dtype=torch.half, device=dev), torch.empty((I, H), dtype=torch.half, device=dev),
        -1, -1, moe.g_B, moe.g_suh, moe.g_svh, K, False, True, moe.u_B, moe.u_suh, moe.u_svh, K, False, True, moe.d_B, moe.d_suh, moe.d_svh, K, False, True,
        True, False, False, sh_bc, sh_gate, 0.0, [bcl(l) for l in moe.gate], [bcl(l) for l in moe.up], [bcl(l) for l in moe.down],
        None, None, None, torch.empty((bszn, H), dtype=torch.half, device=dev))

    def mlp(gl, ul, dl, xr):
        g = _lin(gl, xr).astype(np.float32); u = _lin(ul, xr).astype(np.float32)
        return _lin(dl, (g / (1 + np.exp(-g)) * u).astype(np.float16), out_fp32=True)

    for bsz in (1, 3):
        x = rng.standard_normal((bsz, H)).astype(np.float16)
        sel = np.stack([rng.permutation(E)[:top] for _ in range(bsz)]).astype(np.int64)
        wts = rng.uniform(0.2, 0.8, (bsz, top)).astype(np.float16)
        bc.run_bszN(T(x), T(sel), T(wts))
        ref = np.zeros((bsz, H), np.float32)
        for t in range(bsz):
            for j in range(top):
                e = int(sel[t, j])
                ref[t] += np.float32(wts[t, j]) * mlp(moe.gate[e], moe.up[e], moe.down[e], x[t:t + 1])[0]
        shared = mlp(sh.gate[0], sh.up[0], sh.down[0], x)
        ref = o.add_sigmoid_gate_proj(shared, x, ref, gw) if gate else ref + shared
        got = to_[:bsz].cpu().numpy()
        assert np.abs(got - ref).max() / np.sqrt((ref ** 2).mean()) < 2e-2, bsz
    if gate:
        return
    # BC_MLP: up -> gelu (against ones) -> down, the model's width 224 padded to the quantized 256
    hs = 224
    up_l, dn_l = moe.up[0], moe.down[0]
    xp = torch.zeros((1, H), dtype=torch.half, device=dev); yp = torch.empty((1, H), dtype=torch.half, device=dev)
    m = ext.BC_MLP(xp=xp, u=torch.empty((1, 1, I), dtype=torch.half, device=dev), ones=torch.ones((1, 1, I), dtype=torch.half, device=dev), yp=yp,
                   act_silu=False, act_gelu=True, act_relu2=False, up=bcl(up_l), down=bcl(dn_l), act_limit=0.0, hidden_size=hs, out_size=hs)
    x = rng.standard_normal((1, 1, hs)).astype(np.float16)
    d = torch.zeros((1, 1, hs), dtype=torch.half, device=dev)
    m.run_bsz1(T(x), d)
    xpad = np.zeros((1, H), np.float16); xpad[:, :hs] = x.reshape(1, hs)
    u = _lin(up_l, xpad).astype(np.float16)
    a = o.act_mul(u, np.ones_like(u), "gelu", 0.0)
    ref = _lin(dn_l, a).astype(np.float32)[:, :hs]
    assert np.abs(d.float().cpu().numpy().reshape(1, hs) - ref).max() / np.sqrt((ref ** 2).mean()) < 2e-2


def test_exl3_moe_is_capturable_and_replays_bit_for_bit_with_other_assignments(dev):
    """VERDICT round 2, task 6: ext.exl3_moe has no host round trip (the slot list is built on the device, the scatter runs in a fixed order), so a
    hipGraph captured with ONE routing result replays correctly -- bit for bit against the eager op -- after expert_count / token_sorted /
    weight_sorted / the hidden state have been overwritten in place with OTHER ragged assignments (bsz 8, top-2; idle experts, an over-limit
    expert, counts that change the number of used slots)."""
    from exllamav3_amd import ext
    from exllamav3_amd.moe_path import SyntheticEXL3MoE
    hidden, inter, E, bsz, top_k, max_rows = 256, 384, 6, 8, 2, 5
    moe = SyntheticEXL3MoE(hidden, inter, experts=E, top_k=top_k, K=4, cb=2, device=dev, seed=11)
    rng = np.random.default_rng(8)
    Tn = bsz * top_k

    def routing(seed):
        r = np.random.default_rng(seed)
        sel = np.stack([r.permutation(E)[:top_k] for _ in range(bsz)])                   # (bsz, top_k) distinct experts per token
        if seed == 2: sel[:, 0] = 3; sel[:, 1] = np.where(sel[:, 1] == 3, 0, sel[:, 1])  # expert 3 gets all 8 tokens: above max_rows, skipped
        w = r.uniform(0.1, 0.9, (bsz, top_k)).astype(np.float16)
        order = np.argsort(sel.reshape(-1), kind="stable")
        counts = np.bincount(sel.reshape(-1), minlength=E)
        return (np.concatenate([counts, [0]]).astype(np.int64), (order // top_k).astype(np.int64), w.reshape(-1)[order],
                r.standard_normal((bsz, hidden)).astype(np.float16))

    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    c0, t0, w0, x0 = routing(1)
    cnt, tok, wts, x = T(c0), T(t0), T(w0), T(x0)
    out = torch.zeros((bsz, hidden), dtype=torch.float, device=dev)
    tmp_s = torch.empty((4, max_rows, hidden), dtype=torch.half, device=dev); tmp_i = torch.empty((4, max_rows, inter), dtype=torch.half, device=dev)
    args = (x, out, cnt, tok, wts, tmp_s, tmp_s, tmp_i, tmp_i, ext.MOE_ACT_SILU, 4, 4, 4, moe.g_B, moe.g_suh, moe.g_svh, moe.u_B, moe.u_suh, moe.u_svh,
            moe.d_B, moe.d_suh, moe.d_svh, False, True, False, True, False, True, 0.0, 3)
    ext.exl3_moe(*args); torch.cuda.synchronize()                                       # warm-up outside capture
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            ext.exl3_moe(*args)
    torch.cuda.synchronize()
    seen = []
    for seed in (1, 2, 3, 4):
        c, t, w, xv = routing(seed)
        cnt.copy_(T(c)); tok.copy_(T(t)); wts.copy_(T(w)); x.copy_(T(xv))
        base = T(rng.standard_normal((bsz, hidden)).astype(np.float32))
        out.copy_(base); ext.exl3_moe(*args); torch.cuda.synchronize()
        eager = out.clone()
        out.copy_(base); g.replay(); torch.cuda.synchronize()
        assert torch.equal(out, eager), f"replay differs from eager (routing {seed})"
        out.copy_(base); g.replay(); torch.cuda.synchronize()
        assert torch.equal(out, eager)
        assert bool((eager != base).any())
        seen.append(eager.clone())
        if seed == 2:
            # expert 3 (8 tokens > max_rows) contributes nothing, every token still got its other expert's output
            assert bool((eager != base).any(dim=1).all())
    assert not torch.equal(seen[0], seen[2])


@pytest.mark.parametrize("experts,hidden", [(8, 4096), (6, 384), (64, 1024)])
@pytest.mark.parametrize("tokens", [1, 3])
def test_router_with_rmsnorm_inside_matches_norm_then_router(dev, experts, hidden, tokens):
    """routing_std_norm (RMSNorm of the residual formed inside the router launch, mean square from glue_resid's per-block sums) against the
    oracle's rms_norm -> routing_std and against the two separate HIP launches: normalised rows to an fp16 ulp (the mean square is summed in
    another order), same experts, weights within fp16; Mixtral's 4096 x 8, a non-multiple-of-8 expert count (slow path) and 64 experts."""
    from exllamav3_amd import ext
    rng = np.random.default_rng(experts + tokens)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    x = (rng.standard_normal((tokens, hidden)) * 3).astype(np.float16)
    w = (1 + 0.1 * rng.standard_normal(hidden)).astype(np.float16)
    gate = (rng.standard_normal((hidden, experts)) / np.sqrt(hidden)).astype(np.float16)
    K = 2
    dx = T(x)
    ss = torch.empty((tokens, hidden // 128), dtype=torch.float, device=dev)
    ext.glue_resid(None, 0, None, None, dx, ss, tokens)                       # per-block sums of squares of x (x itself unchanged)
    xn = torch.empty_like(dx); sc = torch.empty((tokens, experts), dtype=torch.half, device=dev)
    sel = torch.empty((tokens, K), dtype=torch.long, device=dev); wt = torch.empty((tokens, K), dtype=torch.half, device=dev)
    slots = torch.empty((2, tokens * K), dtype=torch.long, device=dev)
    ext.routing_std_norm(dx, T(w), ss, 1e-5, xn, T(gate), sc, sel, wt, gu_slots=slots)
    xn_ref = o.rms_norm(x, w, 1e-5)
    got = xn.float().cpu().numpy()
    assert np.abs(got - xn_ref.astype(np.float32)).max() <= 2e-3 * np.abs(xn_ref.astype(np.float32)).max()
    # route the oracle on the kernel's own xn (an ulp in xn may flip a near-tie otherwise)
    scores, sel_ref, w_ref = o.routing_std(xn.cpu().numpy(), gate, K)
    assert np.abs(sc.float().cpu().numpy() - scores.astype(np.float32)).max() < 2e-3 * max(1.0, float(np.abs(scores).max()))
    assert np.array_equal(sel.cpu().numpy(), sel_ref)
    assert np.abs(wt.float().cpu().numpy() - w_ref.astype(np.float32)).max() < 2e-3
    assert np.array_equal(slots.cpu().numpy(), np.stack([sel_ref.reshape(-1), sel_ref.reshape(-1) + experts]))
    # the two separate launches on the same xn give the same bits
    sc2 = torch.empty_like(sc); sel2 = torch.empty_like(sel); wt2 = torch.empty_like(wt)
    ext.routing_std(xn, T(gate), sc2, sel2, wt2)
    assert torch.equal(sc, sc2) and torch.equal(sel, sel2) and torch.equal(wt, wt2)
    # per_expert_scale (bf16): the selected experts' softmax weights times their scale
    pes = torch.from_numpy(rng.uniform(0.5, 2.0, experts).astype(np.float32)).to(torch.bfloat16)
    ext.routing_std(xn, T(gate), sc2, sel2, wt2, per_expert_scale=pes.to(dev))
    _, _, w_s = o.routing_std(xn.cpu().numpy(), gate, K, per_expert_scale=pes.float().numpy())
    assert torch.equal(sel2, sel) and np.abs(wt2.float().cpu().numpy() - w_s.astype(np.float32)).max() < 4e-3


def test_fused_step_with_attention_matches_oracle(dev):
    """decode_step_fused(with_attention): q/k/v -> RoPE -> quantized append -> attention over the (pre-filled) quantized cache -> o_proj ...
    against the oracle composition (attention over the dequantized cache)."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    shape = LlamaShape("tiny", 256, 512, 1, 4, 2, 128, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=1024)
    bsz, pos = 2, 300
    model.alloc_state(bsz, pos=pos)
    model.with_attention = True
    # pre-fill the cache of layer 0 with random context (contiguous pages per sequence in this model)
    rng = np.random.default_rng(3)
    ctx_k = rng.standard_normal((bsz, 1024, model.hkv * 128)).astype(np.float16); ctx_v = rng.standard_normal((bsz, 1024, model.hkv * 128)).astype(np.float16)
    kq, ks = o.kv_quant(ctx_k, 4); vq, vs = o.kv_quant(ctx_v, 4)
    kc, ksc = model.kcache[0]; vc, vsc = model.vcache[0]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    kc.copy_(T(kq.view(np.int32)).view(kc.shape)); ksc.copy_(T(ks).view(ksc.shape)); vc.copy_(T(vq.view(np.int32)).view(vc.shape)); vsc.copy_(T(vs).view(vsc.shape))
    logits = model.decode_step_fused().float().cpu().numpy()
    # oracle
    x = _np(model.x0); L = model.layers[0]
    xn = o.rms_norm(x, _np(L["norm1"]), model.eps)
    q, k, v = _lin(L["q"], xn), _lin(L["k"], xn), _lin(L["v"], xn)
    posv = _np(model.positions)
    q4, k4 = o.rope(q.reshape(bsz, 1, model.hq, 128), k.reshape(bsz, 1, model.hkv, 128), _np(model.inv_freq), positions=posv, rope_mode=o.ROPE_NEOX)
    knq, kns = o.kv_quant(k4.reshape(bsz, 1, -1), 4); vnq, vns = o.kv_quant(v.reshape(bsz, 1, -1), 4)
    kq[:, pos:pos + 1] = knq; ks[:, pos:pos + 1] = kns; vq[:, pos:pos + 1] = vnq; vs[:, pos:pos + 1] = vns
    kd = o.kv_dequant(kq, ks, 4).reshape(bsz, 1024, model.hkv, 128); vd = o.kv_dequant(vq, vs, 4).reshape(bsz, 1024, model.hkv, 128)
    att = o.attn_decode_qcache(q4.reshape(bsz, model.hq, 128), kd, vd, [pos + 1] * bsz)
    ov = _lin(L["o"], att.reshape(bsz, -1), out_fp32=True)
    xn, x = o.rms_norm(ov, _np(L["norm2"]), model.eps, residual_in=x)
    gf, uf = _lin(L["gate"], xn).astype(np.float32), _lin(L["up"], xn).astype(np.float32)
    a = (gf / (1 + np.exp(-gf)) * uf).astype(np.float16)
    d = _lin(L["down"], a, out_fp32=True)
    xn, x = o.rms_norm(d, _np(model.final_norm), model.eps, residual_in=x)
    ref = _lin(model.lm_head, xn).astype(np.float32)
    assert np.abs(logits - ref).max() / np.sqrt((ref ** 2).mean()) < 3e-2


@pytest.mark.parametrize("pos", [3, 70, 300, 1000, 3000])
@pytest.mark.parametrize("bsz", [1, 2])
def test_attention_merge_inside_oproj_is_bit_identical_to_the_merge_launch(dev, pos, bsz):
    """fx step with the decode attention in it (head_dim 128): the flash-decoding merge of the context splits as o_proj's preparation task
    (ext.attn_decode_qcache_split + ext.exl3_gemv_ex_attm, round 4) against attn_decode_qcache (split + merge launches) + o_proj: logits, final residual
    and the K / V rows appended -- bit for bit up to 32 splits of 64 tokens, from a 4-token context (one split: the half-wave kernel writes a record
    instead of its output); at 3001 tokens the fused form takes longer splits (another summation order: a few fp16 ulps of the logits, 1e-2 bound); graph replay reproduces the eager bits;
    the oracle bound holds."""
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    shape = LlamaShape("tiny", 512, 1024, 2, 4, 2, 128, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=4096)
    model.alloc_state(bsz, pos=pos)
    model.with_attention = True
    rng = np.random.default_rng(pos)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ctx = []
    for li in range(shape.layers):
        ck = rng.standard_normal((bsz, 4096, model.hkv * 128)).astype(np.float16); cv = rng.standard_normal((bsz, 4096, model.hkv * 128)).astype(np.float16)
        kq, ks = o.kv_quant(ck, 4); vq, vs = o.kv_quant(cv, 4)
        kc, ksc = model.kcache[li]; vc, vsc = model.vcache[li]
        kc.copy_(T(kq.view(np.int32)).view(kc.shape)); ksc.copy_(T(ks).view(ksc.shape)); vc.copy_(T(vq.view(np.int32)).view(vc.shape)); vsc.copy_(T(vs).view(vsc.shape))
        ctx.append((kq, ks, vq, vs))
    saved_cache = [(c.clone(), s_.clone()) for c, s_ in model.kcache + model.vcache]
    outs = []
    # (merge inside o_proj, q|k|v epilogue inside the context-split launch): the three forms of the attention sublayer -- 4 / 3 / 2 launches around o_proj
    for fused, qkv_in_split in ((False, False), (True, False), (True, True)):
        model.attn_merge_in_oproj, model.attn_qkv_in_split = fused, qkv_in_split
        for (c, s_), (c0, s0) in zip(model.kcache + model.vcache, saved_cache):
            c.copy_(c0); s_.copy_(s0)                                     # every form appends to the same pre-filled cache
        lg = model.decode_step_fx().float().cpu().numpy().copy()
        outs.append((lg, model.x_final.clone(), [(c.clone(), s_.clone()) for c, s_ in model.kcache + model.vcache]))
    (l0, x0_, kv0), (l1, x1_, kv1), (l2, x2_, kv2) = outs
    # the one-launch form (round 4: ext.attn_decode_qcache_split_qkv) against the two launches it replaces: logits, residual and EVERY cache word and scale
    # of every layer, at every context length (same splits, shared device functions; below the matrix-pipe kernel's 128-token bound it falls back)
    assert np.array_equal(l1, l2) and torch.equal(x1_, x2_)
    assert all(torch.equal(a1, a2) and torch.equal(b1, b2) for (a1, b1), (a2, b2) in zip(kv1, kv2))
    l1, x1_, kv1 = l2, x2_, kv2
    assert np.isfinite(l1).all()
    if pos <= 1000:
        assert np.array_equal(l0, l1) and torch.equal(x0_, x1_)
    else:
        # beyond 32 splits of 64 tokens the fused form takes fewer, longer splits than the merge launch would (its statistics live in one chunk of 32
        # lanes): the same attention, another summation order
        assert np.abs(l0 - l1).max() / np.sqrt((l0 ** 2).mean()) < 1e-2
    assert all(torch.equal(a0, a1) and torch.equal(b0, b1) for (a0, b0), (a1, b1) in zip(kv0[:2 * shape.layers][:1], kv1[:2 * shape.layers][:1]))      # layer 0's K rows: same inputs
    # oracle (attention over the dequantized cache incl. the appended token)
    x = _np(model.x0); pend = None
    for li, L in enumerate(model.layers):
        if pend is None: xn = o.rms_norm(x, _np(L["norm1"]), model.eps)
        else: xn, x = o.rms_norm(pend, _np(L["norm1"]), model.eps, residual_in=x)
        q, k, v = _lin(L["q"], xn), _lin(L["k"], xn), _lin(L["v"], xn)
        q4, k4 = o.rope(q.reshape(bsz, 1, model.hq, 128), k.reshape(bsz, 1, model.hkv, 128), _np(model.inv_freq), positions=_np(model.positions), rope_mode=o.ROPE_NEOX)
        kq, ks, vq, vs = ctx[li]
        knq, kns = o.kv_quant(k4.reshape(bsz, 1, -1), 4); vnq, vns = o.kv_quant(v.reshape(bsz, 1, -1), 4)
        kq[:, pos:pos + 1] = knq; ks[:, pos:pos + 1] = kns; vq[:, pos:pos + 1] = vnq; vs[:, pos:pos + 1] = vns
        kd = o.kv_dequant(kq[:, :pos + 1], ks[:, :pos + 1], 4).reshape(bsz, pos + 1, model.hkv, 128); vd = o.kv_dequant(vq[:, :pos + 1], vs[:, :pos + 1], 4).reshape(bsz, pos + 1, model.hkv, 128)
        att = o.attn_decode_qcache(q4.reshape(bsz, model.hq, 128), kd, vd, [pos + 1] * bsz).reshape(bsz, -1)
        ov = _lin(L["o"], att, out_fp32=True)
        xn, x = o.rms_norm(ov, _np(L["norm2"]), model.eps, residual_in=x)
        gf, uf = _lin(L["gate"], xn).astype(np.float32), _lin(L["up"], xn).astype(np.float32)
        a = (gf / (1 + np.exp(-gf)) * uf).astype(np.float16)
        pend = _lin(L["down"], a, out_fp32=True)
    xn, x = o.rms_norm(pend, _np(model.final_norm), model.eps, residual_in=x)
    ref = _lin(model.lm_head, xn).astype(np.float32)
    assert np.abs(l1 - ref).max() / np.sqrt((ref ** 2).mean()) < 3e-2
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            model.decode_step_fx()
    model.logits.zero_(); g.replay(); torch.cuda.synchronize()
    assert np.array_equal(model.logits.float().cpu().numpy(), l1)


@pytest.mark.parametrize("hq,hkv", [(8, 2), (16, 8), (4, 4)])
@pytest.mark.parametrize("pos,bsz", [(3, 1), (70, 2), (300, 1), (500, 2), (1000, 1)])
def test_attention_merge_inside_oproj_head_dim_64(dev, hq, hkv, pos, bsz):
    """The flash-decoding merge as o_proj's preparation task at head_dim 64 (Llama-3.2-1B's heads): a Hadamard block of o_proj's input holds TWO query
    heads whose partial records live in different (kv block, query index) records -- per-lane record addressing, 16-lane statistics, at most 16
    splits.  Against attn_decode_qcache (split + merge launches) + o_proj: bit for bit while both forms take the same splits (<= 512 tokens), 1e-2
    beyond (the merge launch takes 32 shorter splits there)."""
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    shape = LlamaShape("tiny", 512, 768, 2, hq, hkv, 64, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=2048)
    model.alloc_state(bsz, pos=pos)
    model.with_attention = True
    g = torch.Generator(device="cpu").manual_seed(hq * 1000 + pos)
    for c, s_ in model.kcache + model.vcache:
        c.copy_(torch.randint(-2 ** 31, 2 ** 31 - 1, c.shape, generator=g, dtype=torch.int64).to(torch.int32).to(dev))
        s_.copy_((torch.rand(s_.shape, generator=g) * 0.5 + 0.05).half().to(dev))
    saved = [(c.clone(), s_.clone()) for c, s_ in model.kcache + model.vcache]
    outs = []
    model.attn_merge_in_oproj_hd64 = True                                 # (None = only where the matrix-pipe split kernel applies: llama_path.py)
    for fused in (False, True):
        model.attn_merge_in_oproj = fused
        for (c, s_), (c0, s0) in zip(model.kcache + model.vcache, saved):
            c.copy_(c0); s_.copy_(s0)
        lg = model.decode_step_fx().float().cpu().numpy().copy()
        outs.append((lg, model.x_final.clone()))
    (l0, x0_), (l1, x1_) = outs
    assert np.isfinite(l1).all()
    if pos <= 511:
        assert np.array_equal(l0, l1) and torch.equal(x0_, x1_)
    else:
        assert np.abs(l0 - l1).max() / np.sqrt((l0 ** 2).mean()) < 1e-2


@pytest.mark.parametrize("hq,hkv,hd", [(4, 2, 128), (8, 1, 128), (8, 2, 64), (16, 8, 64)])
@pytest.mark.parametrize("bsz", [1, 2])
def test_qkv_epilogue_inside_oproj_is_bit_identical_to_the_glue_launch(dev, hq, hkv, hd, bsz):
    """fx step WITHOUT the attention core (o_proj's input is the finished q): the q|k|v epilogue inside o_proj's launch (ext.exl3_gemv_ex_qkvm: preparation
    tasks finish the q block they need, idle half-waves of the column-block-0 workgroups append K / V) against glue_qkv_rs + o_proj as two launches --
    logits, final residual, the last layer's q and every cache word and scale of every layer bit for bit; head_dim 128 and 64 (two heads per
    Hadamard block), batch 1 and 2 (the second row's K / V tasks land on busy half-waves); graph replay reproduces the eager bits."""
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    shape = LlamaShape("tiny", 512, 768, 2, hq, hkv, hd, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=1024)
    model.alloc_state(bsz, pos=77)
    g = torch.Generator(device="cpu").manual_seed(hq * 100 + hd + bsz)
    for c, s_ in model.kcache + model.vcache:
        c.copy_(torch.randint(-2 ** 31, 2 ** 31 - 1, c.shape, generator=g, dtype=torch.int64).to(torch.int32).to(dev))
        s_.copy_((torch.rand(s_.shape, generator=g) * 0.5 + 0.05).half().to(dev))
    saved = [(c.clone(), s_.clone()) for c, s_ in model.kcache + model.vcache]
    outs = []
    for fused in (False, True):
        model.qkv_in_oproj = fused
        for (c, s_), (c0, s0) in zip(model.kcache + model.vcache, saved):
            c.copy_(c0); s_.copy_(s0)
        model.q.zero_()
        lg = model.decode_step_fx().float().cpu().numpy().copy()
        outs.append((lg, model.x_final.clone(), model.q.clone(), [(c.clone(), s_.clone()) for c, s_ in model.kcache + model.vcache]))
    (l0, x0_, q0, kv0), (l1, x1_, q1, kv1) = outs
    assert np.isfinite(l1).all()
    assert np.array_equal(l0, l1) and torch.equal(x0_, x1_) and torch.equal(q0, q1)
    assert all(torch.equal(a0, a1) and torch.equal(b0, b1) for (a0, b0), (a1, b1) in zip(kv0, kv1))
    assert all(not torch.equal(a1, a0) for (a1, _), (a0, _) in zip(kv1, saved))                  # something was appended in every layer
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            model.decode_step_fx()
    model.logits.zero_(); gr.replay(); torch.cuda.synchronize()
    assert np.array_equal(model.logits.float().cpu().numpy(), l1)


@pytest.mark.parametrize("gq", [1, 2, 3, 5, 6, 7, 8])
@pytest.mark.parametrize("pos,bsz", [(130, 1), (1000, 2)])
def test_attention_qkv_in_split_every_group_size(dev, gq, pos, bsz):
    """ext.attn_decode_qcache_split_qkv for every GQA group size the kernel is instantiated for (the K / V tasks sit on half-waves (GQ) % 8 and
    (GQ + 1) % 8: beside the query tasks up to GQ = 6, in a second task round for 7 and 8): fx step with the q|k|v epilogue inside the context-split
    launch against glue_qkv_rs + split as two launches -- logits, residual, q and every cache word bit for bit, random pre-filled cache."""
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    shape = LlamaShape("tiny", 512, 768, 2, gq, 1, 128, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=2048)
    model.alloc_state(bsz, pos=pos)
    model.with_attention = True
    g = torch.Generator(device="cpu").manual_seed(gq * 1000 + pos)
    for c, s_ in model.kcache + model.vcache:
        c.copy_(torch.randint(-2 ** 31, 2 ** 31 - 1, c.shape, generator=g, dtype=torch.int64).to(torch.int32).to(dev))
        s_.copy_((torch.rand(s_.shape, generator=g) * 0.5 + 0.05).half().to(dev))
    saved = [(c.clone(), s_.clone()) for c, s_ in model.kcache + model.vcache]
    outs = []
    for qkv_in_split in (False, True):
        model.attn_qkv_in_split = qkv_in_split
        for (c, s_), (c0, s0) in zip(model.kcache + model.vcache, saved):
            c.copy_(c0); s_.copy_(s0)
        model.q.zero_()
        lg = model.decode_step_fx().float().cpu().numpy().copy()
        outs.append((lg, model.x_final.clone(), model.q.clone(), [(c.clone(), s_.clone()) for c, s_ in model.kcache + model.vcache]))
    (l0, x0_, q0, kv0), (l1, x1_, q1, kv1) = outs
    assert np.isfinite(l1).all()
    assert np.array_equal(l0, l1) and torch.equal(x0_, x1_) and torch.equal(q0, q1)      # q: the last layer's finished queries (split 0 writes them)
    assert all(torch.equal(a0, a1) and torch.equal(b0, b1) for (a0, b0), (a1, b1) in zip(kv0, kv1))
    # something was appended: the new token's row differs from the pre-filled words in every layer
    assert all(not torch.equal(a1, a0) for (a1, _), (a0, _) in zip(kv1, saved))


@pytest.mark.parametrize("waves", [8])
@pytest.mark.parametrize("hd,hq,hkv", [(128, 4, 1), (128, 7, 1), (128, 8, 1), (64, 8, 2), (64, 16, 8)])
@pytest.mark.parametrize("pos,bsz", [(1000, 2), (1900, 1)])
def test_attention_qkv_in_split_eight_waves(dev, hd, hq, hkv, pos, bsz, waves):
    """The matrix-pipe split kernel with 8 waves per workgroup (round 5: two token-step chains per SIMD; the default from two 64-token steps
    per split on) in its FUSED form -- 16 half-waves share the q|k|v epilogue's tasks in ONE round -- against glue_qkv_rs + the split launch with the
    same wave count: logits, residual, q, every cache word and scale bit for bit; and against the 4-wave form within the merge's rounding (another
    summation order of the same partial sums)."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    shape = LlamaShape("tiny", 512, 768, 2, hq, hkv, hd, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=2048)
    model.alloc_state(bsz, pos=pos)
    model.with_attention = True
    model.attn_merge_in_oproj_hd64 = True
    g = torch.Generator(device="cpu").manual_seed(hq * 1000 + pos)
    for c, s_ in model.kcache + model.vcache:
        c.copy_(torch.randint(-2 ** 31, 2 ** 31 - 1, c.shape, generator=g, dtype=torch.int64).to(torch.int32).to(dev))
        s_.copy_((torch.rand(s_.shape, generator=g) * 0.5 + 0.05).half().to(dev))
    saved = [(c.clone(), s_.clone()) for c, s_ in model.kcache + model.vcache]
    outs = []
    try:
        for w, qkv_in_split in ((waves, False), (waves, True), (4, True)):
            ext.set_attn_wide_waves(w)
            model.attn_qkv_in_split = qkv_in_split
            for (c, s_), (c0, s0) in zip(model.kcache + model.vcache, saved):
                c.copy_(c0); s_.copy_(s0)
            model.q.zero_()
            lg = model.decode_step_fx().float().cpu().numpy().copy()
            outs.append((lg, model.x_final.clone(), model.q.clone(), [(c.clone(), s_.clone()) for c, s_ in model.kcache + model.vcache]))
    finally:
        ext.set_attn_wide_waves(0)
    (l0, x0_, q0, kv0), (l1, x1_, q1, kv1), (l4, _, q4, kv4) = outs
    assert np.isfinite(l1).all()
    assert np.array_equal(l0, l1) and torch.equal(x0_, x1_) and torch.equal(q0, q1)
    assert all(torch.equal(a0, a1) and torch.equal(b0, b1) for (a0, b0), (a1, b1) in zip(kv0, kv1))
    assert all(not torch.equal(a1, a0) for (a1, _), (a0, _) in zip(kv1, saved))
    # layer 0's new K / V rows do not depend on the attention output: the same words whatever the wave count
    assert torch.equal(kv1[0][0], kv4[0][0]) and torch.equal(kv1[2][0], kv4[2][0])             # (kcache + vcache of two layers: K0, K1, V0, V1)
    assert np.abs(l1 - l4).max() / np.sqrt((l4 ** 2).mean()) < 1e-2


@pytest.mark.parametrize("hq,hkv", [(8, 2), (16, 8), (4, 4), (6, 2), (16, 2)])
@pytest.mark.parametrize("pos,bsz", [(70, 1), (130, 1), (500, 2), (1000, 2)])
def test_attention_qkv_in_split_head_dim_64(dev, hq, hkv, pos, bsz):
    """ext.attn_decode_qcache_split_qkv at head_dim 64 (round 4): the matrix-pipe split kernel on the two kv heads of a 128-value block with the q|k|v
    epilogue in its preparation phase -- tasks are 128-value blocks (two query heads each; the K and the V block of both kv heads), the rope partner
    distance is 8 lanes.  Against glue_qkv_rs + the split launch (merge inside o_proj in both): logits, residual, q and every cache word and scale bit
    for bit; group sizes 1 .. 4 take the one-launch form, 8 (two launches inside the entry point) and the 70-token context (below the kernel's bound) fall back."""
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    shape = LlamaShape("tiny", 512, 768, 2, hq, hkv, 64, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=2048)
    model.alloc_state(bsz, pos=pos)
    model.with_attention = True
    model.attn_merge_in_oproj_hd64 = True
    g = torch.Generator(device="cpu").manual_seed(hq * 1000 + pos)
    for c, s_ in model.kcache + model.vcache:
        c.copy_(torch.randint(-2 ** 31, 2 ** 31 - 1, c.shape, generator=g, dtype=torch.int64).to(torch.int32).to(dev))
        s_.copy_((torch.rand(s_.shape, generator=g) * 0.5 + 0.05).half().to(dev))
    saved = [(c.clone(), s_.clone()) for c, s_ in model.kcache + model.vcache]
    outs = []
    for qkv_in_split in (False, True):
        model.attn_qkv_in_split = qkv_in_split
        for (c, s_), (c0, s0) in zip(model.kcache + model.vcache, saved):
            c.copy_(c0); s_.copy_(s0)
        model.q.zero_()
        lg = model.decode_step_fx().float().cpu().numpy().copy()
        outs.append((lg, model.x_final.clone(), model.q.clone(), [(c.clone(), s_.clone()) for c, s_ in model.kcache + model.vcache]))
    (l0, x0_, q0, kv0), (l1, x1_, q1, kv1) = outs
    assert np.isfinite(l1).all()
    assert np.array_equal(l0, l1) and torch.equal(x0_, x1_) and torch.equal(q0, q1)
    assert all(torch.equal(a0, a1) and torch.equal(b0, b1) for (a0, b0), (a1, b1) in zip(kv0, kv1))
    assert all(not torch.equal(a1, a0) for (a1, _), (a0, _) in zip(kv1, saved))


@pytest.mark.parametrize("bsz", [1, 16])
def test_fused_pipeline_on_70b_tp8_rank_shapes(dev, bsz):
    """The per-rank shapes of Llama-3.1-70B under TP = 8 (hidden 8192, q 8 heads, ONE kv head -> 128-column k / v matrices, inter 3584,
    vocab shard 16128): fused pipeline against the op-by-op pipeline (single process; the collectives are exercised by test_tp_gloo.py)."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    shape = LlamaShape("70b-rank", 8192, 3584, 1, 8, 1, 128, 16128)
    model = SyntheticEXL3Llama(shape, K=3, cb=2, device=dev, kv_bits=4, max_ctx=1024)
    model.alloc_state(bsz, pos=500)
    lu = model.decode_step().float().cpu().numpy().copy()
    lf = model.decode_step_fused().float().cpu().numpy()
    assert np.isfinite(lf).all()
    assert np.abs(lf - lu).max() / np.sqrt((lu ** 2).mean()) < 1e-2


def test_checkpoint_round_trip_reproduces_logits(dev, tmp_path):
    """save_checkpoint -> from_checkpoint (safetensors + config.json, HF Llama naming): the loaded model owns bit-identical tensors and the
    fused decode step gives bit-identical logits; per-group bitrates may differ (o / down / lm_head at other K than q|k|v and gate|up)."""
    from exllamav3_amd import linear
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama, _rand_linear
    shape = LlamaShape("tiny", 256, 512, 2, 4, 2, 128, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=2048, head_K=6)
    g = torch.Generator(device=dev); g.manual_seed(7)
    for L in model.layers:                                                    # mixed bitrates across groups
        L["o"] = _rand_linear(512, 256, 3, 0, dev, g, out_dtype=torch.float)
        L["down"] = _rand_linear(512, 256, 5, 1, dev, g, out_dtype=torch.float)
    model.save_checkpoint(str(tmp_path))
    loaded = SyntheticEXL3Llama.from_checkpoint(str(tmp_path), device=dev, max_ctx=2048)
    assert (loaded.K, loaded.cb, loaded.lm_head.K, loaded.layers[0]["o"].K, loaded.layers[1]["down"].mcg) == (4, 2, 6, 3, True)
    assert loaded.layers[0]["o"].in_features == 512
    for La, Lb in zip(model.layers, loaded.layers):
        for nm in ("q", "k", "v", "o", "gate", "up", "down"):
            assert torch.equal(La[nm].trellis, Lb[nm].trellis) and torch.equal(La[nm].suh, Lb[nm].suh) and torch.equal(La[nm].svh, Lb[nm].svh)
        assert torch.equal(La["norm1"], Lb["norm1"]) and torch.equal(La["norm2"], Lb["norm2"])
    for bsz in (1, 6):
        model.alloc_state(bsz, pos=77); loaded.alloc_state(bsz, pos=77)
        loaded.x0.copy_(model.x0)
        la = model.decode_step_fused().clone()
        lb = loaded.decode_step_fused()
        assert torch.equal(la, lb)
    # a q|k|v or gate|up group that mixes bitrates / codebooks runs as one GEMV per matrix (the reference's fallback, modules/attn.py:439,
    # modules/mlp.py:635): same logits as the op-by-op pipeline up to fp32 summation order, at m <= 4 and above
    model.layers[0]["k"] = _rand_linear(256, 256, 5, 2, dev, g)
    model.layers[1]["up"] = _rand_linear(256, 512, 3, 0, dev, g)
    model.save_checkpoint(str(tmp_path / "mixed"))
    mixed = SyntheticEXL3Llama.from_checkpoint(str(tmp_path / "mixed"), device=dev, max_ctx=2048)
    assert (mixed.layers[0]["k"].K, mixed.layers[0]["q"].K, mixed.layers[1]["up"].K, mixed.layers[1]["gate"].K) == (5, 4, 3, 4)
    for bsz in (1, 3, 6):
        mixed.alloc_state(bsz, pos=90)
        lu = mixed.decode_step().float().cpu().numpy().copy()
        for c, s in mixed.kcache + mixed.vcache:
            c.zero_(); s.zero_()
        lf = mixed.decode_step_fused().float().cpu().numpy()
        assert np.isfinite(lf).all()
        assert np.abs(lf - lu).max() / np.sqrt((lu ** 2).mean()) < 1e-2
    mixed.prefill_chunk(160)                                                  # the prefill route reconstructs every matrix with its own K
    torch.cuda.synchronize()


@pytest.mark.parametrize("tokens", [1, 3])
def test_moe_expert_parallel_partials_sum_to_the_whole_block(dev, tokens, tmp_path):
    """Expert parallelism (BASELINE config 5, TP = 2): two blocks loaded from the same checkpoint with expert ranges [0, 3) and [3, 6) return
    partial sums over their local experts (range-filtered indexed launches, quant/exl3_gemm_kernel.cuh:99-127); their sum -- what the
    all-reduce produces -- equals the whole block's output, and the whole block loaded from files equals the original bit for bit."""
    from exllamav3_amd.moe_path import SyntheticEXL3MoE
    moe = SyntheticEXL3MoE(256, 384, experts=6, top_k=2, K=4, cb=2, device=dev, seed=11)
    moe.save_checkpoint(str(tmp_path))
    x = torch.randn((tokens, 256), device=dev, generator=torch.Generator(device=dev).manual_seed(tokens)).half()
    y = moe.forward(x).float().clone()
    whole = SyntheticEXL3MoE.from_checkpoint(str(tmp_path), top_k=2, device=dev)
    assert (whole.hidden, whole.inter, whole.E, whole.K, whole.cb) == (256, 384, 6, 4, 2)
    assert torch.equal(whole.forward(x).float(), y)
    parts = []
    for first, last in ((0, 3), (3, 6)):
        part = SyntheticEXL3MoE.from_checkpoint(str(tmp_path), top_k=2, device=dev, first_expert=first, last_expert=last)
        assert len(part.gate) == 3
        parts.append(part.forward(x).float().clone())
        assert torch.equal(part.sel, moe.sel)                                   # the router is replicated
    s = parts[0] + parts[1]
    assert float((s - y).abs().max()) < 2e-3 * float(y.abs().max()) + 1e-4
    # every token has at least one rank with work; a rank whose range holds none of a token's experts returns zeros for it
    for i in range(tokens):
        local0 = int(((moe.sel[i] >= 0) & (moe.sel[i] < 3)).sum())
        if local0 == 0:
            assert float(parts[0][i].abs().max()) == 0.0


@pytest.mark.parametrize("model_name,tp", [("llama-3.1-8b", 2), ("llama-3.1-8b", 4), ("llama-3.1-8b", 8), ("llama-3.1-70b", 8)])
@pytest.mark.parametrize("bsz", [1, 16])
def test_tp_rank_code_path_on_one_gpu(dev, model_name, tp, bsz):
    """The tensor-parallel branches of the pipelines (non-deferred o / down with fp32 partial outputs, all-reduce hook, glue_resid on the dense
    tensor, sharded lm_head) with the real per-rank shapes of the multi-GPU bench, run as rank 0 of a world of `tp` on ONE GPU: the backend's
    all-reduce is replaced by a no-op (the collective itself is covered by test_tp_gloo.py), so fused and op-by-op pipelines both work on
    this rank's partial sums and must agree; one layer, finite logits of the rank's vocabulary shard."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama

    from exllamav3_amd.tp import OneRankOfMany                          # no-op collectives: one rank's compute leg (bench.py uses the same stand-in)

    be = OneRankOfMany(tp, dev)
    model = SyntheticEXL3Llama(SHAPES[model_name], K=4, cb=2, device=dev, backend=be, kv_bits=4, max_ctx=1024, layers=1)
    s = SHAPES[model_name]
    assert (model.hq, model.hkv) == (s.heads_q // tp, s.heads_kv // tp) and model.layers[0]["down"].in_features == model.inter_local
    model.alloc_state(bsz, pos=300)
    lu = model.decode_step().float().cpu().numpy().copy()
    n_unfused = be.calls
    lf = model.decode_step_fused().float().cpu().numpy()
    assert be.calls == 2 * n_unfused and n_unfused == 2                       # o and down of the one layer, in both pipelines
    assert lf.shape == (bsz, model.vocab_local) and np.isfinite(lf).all()
    assert np.abs(lf - lu).max() / np.sqrt((lu ** 2).mean()) < 1e-2
    # graph capture of the rank's step (the bench replays it)
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            model.decode_step_fused()
        g.replay(); st.synchronize()
    assert np.array_equal(model.logits.float().cpu().numpy(), lf)
    if bsz <= model.fx_max_bsz:
        # the TP form of the fx pipeline (round 4: 7 launches per layer, R += the rank's finished rows in the all-reduce launch -- here with the exchange left
        # out) on the same rank shapes, slab route and dense-partials route: the rank's partial-sum logits agree with the glue TP step
        n0 = be.calls
        lx = model.decode_step_fx().float().cpu().numpy().copy()
        assert be.calls == n0 + 2 and np.isfinite(lx).all() and np.abs(lx - lu).max() / np.sqrt((lu ** 2).mean()) < 1.5e-2
        be.ipc_like = False
        ly = model.decode_step_fx().float().cpu().numpy().copy()
        be.ipc_like = True
        assert np.abs(ly - lx).max() / np.sqrt((lu ** 2).mean()) < 1e-2
        assert np.array_equal(model.decode_step_fx().float().cpu().numpy(), lx)


@pytest.mark.parametrize("k,n", [(512, 4096), (1792, 4096), (512, 384), (512, 2048)])
@pytest.mark.parametrize("m", [1, 4, 7])
def test_gemv_resid_tail_equals_gemv_plus_glue_resid(dev, k, n, m):
    """exl3_gemv_resid (o_proj / down_proj with glue_resid inside the launch) against the two-launch route, bit for bit: the residual stream and
    the per-block sums of squares.  n/128 % 8 == 0 takes the XCD-local hand-off (all slices of a column block on one XCD, plain stores + an
    L2 atomic), otherwise -- and with the switch off -- the agent-scope one; 30 repetitions on the same workspace catch stale reads."""
    from exllamav3_amd import ext
    ext.set_gemm3_min_rows(0)
    ext.set_gemv_gen4(False)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rng = np.random.default_rng(k + n + m)
    tr, suh, svh = o.synth_linear(k, n, 4, seed=3, realistic=True)
    ttr, tsu, tsv = T(tr), T(suh), T(svh)
    for rotated in (False, True):
        for rep in range(30 if n == 4096 else 6):
            x = T(rng.standard_normal((m, k)).astype(np.float16))
            r0 = T((rng.standard_normal((m, n)) * 2).astype(np.float16))
            xh = torch.empty_like(x)
            if rotated:
                ext.had_r_128(x, xh, tsu, None, 1.0)
            # reference route: deferred GEMV + glue_resid
            r_ref = r0.clone(); ss_ref = torch.zeros((m, n // 128), dtype=torch.float32, device=dev)
            if rotated:
                slabs, S = ext.exl3_gemv_ex(None, [xh], None, [ttr], None, None, None, m, False, True, ext.GEMV_IN_ROTATED | ext.GEMV_OUT_DEFERRED)
            else:
                slabs, S = ext.exl3_gemv_ex(x, None, None, [ttr], None, [tsu], None, m, False, True, ext.GEMV_OUT_DEFERRED)
            ext.glue_resid(slabs[0], S, tsv, None, r_ref, ss_ref, m)
            for local in (True, False):
                try:
                    ext.set_tail_xcd_local(local)      # opt-in: enabling probes the workgroup -> XCD mapping and refuses if it is not i % 8
                except RuntimeError:
                    assert local
                    continue
                r1 = r0.clone(); ss1 = torch.full((m, n // 128), float("nan"), dtype=torch.float32, device=dev)
                ext.exl3_gemv_resid(None if rotated else x, xh if rotated else None, None, ttr, None if rotated else tsu, tsv, None, m, False, True, r1, ss1,
                                    force_split=S)
                assert torch.equal(r1, r_ref), f"residual differs (rotated={rotated}, local={local}, rep={rep})"
                assert torch.equal(ss1, ss_ref), f"sums of squares differ (rotated={rotated}, local={local}, rep={rep})"
    ext.set_tail_xcd_local(False)


@pytest.mark.parametrize("cb", [0, 2])
@pytest.mark.parametrize("bsz", [1, 3, 4])
@pytest.mark.parametrize("K", [4, 3])
def test_resid_in_gemv_pipeline_matches_oracle_and_glue_pipeline(dev, cb, bsz, K):
    """decode_step_resid (5 launches per layer: the residual add of o_proj / down_proj is finished inside the next q|k|v / gate|up GEMV, RMSNorm
    scale corrected downstream) against the oracle and the glue pipeline; graph replay reproduces the eager bits; the residual stream it
    publishes equals the glue pipeline's to fp16 rounding."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    shape = LlamaShape("tiny", 512, 1024, 3, 4, 2, 128, 384)
    model = SyntheticEXL3Llama(shape, K=K, cb=cb, device=dev, kv_bits=4, max_ctx=2048)
    model.alloc_state(bsz, pos=700)
    lf = model.decode_step_fused().float().cpu().numpy().copy()
    xf = model.x.float().cpu().numpy().copy()
    kf = [(kc.clone(), ks.clone()) for kc, ks in model.kcache]
    for kc, ks in model.kcache + model.vcache:
        kc.zero_(); ks.zero_()
    lr = model.decode_step_resid().float().cpu().numpy().copy()
    assert np.isfinite(lr).all()
    ref = _oracle_decode(model, _np(model.x0))
    rms = np.sqrt((ref ** 2).mean())
    assert np.abs(lr - ref).max() / rms < 3e-2
    assert np.abs(lr - lf).max() / rms < 1.5e-2
    xr = model.x_final.float().cpu().numpy()
    assert np.abs(xr - xf).max() / np.sqrt((xf ** 2).mean()) < 1e-2
    # the quantized K append of every layer landed in the same slots with (nearly) the same scales
    for (kc, ks), (kc0, ks0) in zip(model.kcache, kf):
        assert bool(((ks != 0) == (ks0 != 0)).all())
        assert float((ks.float() - ks0.float()).abs().max()) <= 0.02 * float(ks0.float().abs().max()) + 1e-3
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            model.decode_step_resid()
    for _ in range(3):
        model.logits.zero_(); g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(model.logits.float().cpu().numpy(), lr)


def test_resid_in_gemv_pipeline_with_large_residual_scale_change(dev):
    """The estimate r_prev may be far from r_new (a layer whose output dwarfs the residual): the correction r_new / r_prev keeps the result
    within tolerance as long as fp16(x * w * r_prev) neither overflows nor flushes -- checked with a residual 50x smaller than the sublayer output."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    shape = LlamaShape("tiny", 512, 1024, 2, 4, 2, 128, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=2048)
    model.alloc_state(1, pos=100)
    model.x0.mul_(0.02)
    lr = model.decode_step_resid().float().cpu().numpy().copy()
    ref = _oracle_decode(model, _np(model.x0))
    assert np.abs(lr - ref).max() / np.sqrt((ref ** 2).mean()) < 3e-2


@pytest.mark.parametrize("cb,K", [(2, 4), (0, 3), (1, 5)])
@pytest.mark.parametrize("bsz", [1, 3, 4])
def test_fixed_point_residual_pipeline_matches_oracle_and_glue_pipeline(dev, cb, bsz, K):
    """decode_step_fx (round 3; 6 launches per layer: o_proj / down_proj add their rows into a 64-bit fixed-point residual accumulator with integer
    atomics, q|k|v / gate|up read it and normalise with the previous residual's 1/rms, glue_qkv_rs / glue_act_rs correct) against the oracle and the
    glue pipeline; the residual it leaves equals the glue pipeline's to fp16 rounding; the KV append lands in the same slots; it is bit-reproducible
    across repetitions (integer atomics commute) and under hipGraph replay."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    shape = LlamaShape("tiny", 512, 1024, 3, 4, 2, 128, 384)
    model = SyntheticEXL3Llama(shape, K=K, cb=cb, device=dev, kv_bits=4, max_ctx=2048)
    model.alloc_state(bsz, pos=700)
    model.fx_max_bsz = 4                                     # the kernels take up to 4 rows (the default pipeline choice stops at 2: measured)
    assert model.fx_act_in_gemv                              # the default: 5 launches per layer; the 6-launch form first
    model.fx_act_in_gemv = False
    lf = model.decode_step_fused().float().cpu().numpy().copy()
    xf = model.x.float().cpu().numpy().copy()
    kf = [(kc.clone(), ks.clone()) for kc, ks in model.kcache]
    for kc, ks in model.kcache + model.vcache:
        kc.zero_(); ks.zero_()
    lr = model.decode_step_fx().float().cpu().numpy().copy()
    assert np.isfinite(lr).all()
    ref = _oracle_decode(model, _np(model.x0))
    rms = np.sqrt((ref ** 2).mean())
    assert np.abs(lr - ref).max() / rms < 3e-2
    assert np.abs(lr - lf).max() / rms < 1.5e-2
    xr = model.x_final.float().cpu().numpy()
    assert np.abs(xr - xf).max() / np.sqrt((xf ** 2).mean()) < 1e-2
    for (kc, ks), (kc0, ks0) in zip(model.kcache, kf):
        assert bool(((ks != 0) == (ks0 != 0)).all())
        assert float((ks.float() - ks0.float()).abs().max()) <= 0.02 * float(ks0.float().abs().max()) + 1e-3
    for _ in range(5):                                       # atomics in any arrival order: the same bits every time
        assert np.array_equal(model.decode_step_fx().float().cpu().numpy(), lr)
    # 5 launches per layer: silu(g) * u (row-scale corrected) inside the down launch instead of glue_act_rs
    model.fx_act_in_gemv = True
    la = model.decode_step_fx().float().cpu().numpy().copy()
    assert np.isfinite(la).all() and np.abs(la - ref).max() / rms < 3e-2 and np.abs(la - lr).max() / rms < 1e-2
    assert np.array_equal(model.decode_step_fx().float().cpu().numpy(), la)
    model.fx_act_in_gemv = False
    # 5 launches per layer, the other way: gate|up add into fixed-point accumulators (cleared by the o_proj launch), down forms silu(g) * u from them
    model.fx_gu_atomic = True
    lg_ = model.decode_step_fx().float().cpu().numpy().copy()
    assert np.isfinite(lg_).all() and np.abs(lg_ - ref).max() / rms < 3e-2 and np.abs(lg_ - lr).max() / rms < 1e-2
    for _ in range(3):
        assert np.array_equal(model.decode_step_fx().float().cpu().numpy(), lg_)
    model.fx_gu_atomic = False
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            model.decode_step_fx()
    for _ in range(3):
        model.logits.zero_(); g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(model.logits.float().cpu().numpy(), lr)


def test_fixed_point_residual_pipeline_scale_change_large_values_and_bias(dev):
    """(a) a residual 50x smaller than the sublayer outputs (the 1/rms estimate is far off); (b) residual values in the hundreds (fixed point: no
    overflow, no loss); (c) GEMV_OUT_ATOMIC with a bias on a 3-slice split: the bias is added exactly once."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    shape = LlamaShape("tiny", 512, 1024, 2, 4, 2, 128, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=2048)
    model.alloc_state(1, pos=100)
    for scale in (0.02, 400.0):
        model.x0.copy_(torch.randn_like(model.x0.float()).mul_(scale).half())
        lr = model.decode_step_fx().float().cpu().numpy().copy()
        ref = _oracle_decode(model, _np(model.x0))
        assert np.abs(lr - ref).max() / np.sqrt((ref ** 2).mean()) < 3e-2, scale
    # (c) one linear with bias into an accumulator that already holds a residual
    k, n, m, K = 768, 256, 2, 4
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rng = np.random.default_rng(3)
    tr, su, sv = o.synth_linear(k, n, K, seed=8, realistic=True)
    bias = (rng.standard_normal(n) * 0.3).astype(np.float16)
    x = rng.standard_normal((m, k)).astype(np.float16)
    r0 = (rng.standard_normal((m, n)) * 2).astype(np.float16)
    R = torch.zeros((m, n), dtype=torch.int64, device=dev); ss = torch.empty((m, n // 128), dtype=torch.float32, device=dev)
    ext.fx_init(T(r0), R, ss, m)
    import ctypes
    gemv_ex = ext._lib.lib().exl3_gemv_ex                   # the C entry point: the Python wrapper has no bias argument
    trd, sud, svd, bd, xd = T(tr), T(su), T(sv), T(bias), T(x)
    Bs = (ctypes.c_void_p * 1)(trd.data_ptr()); Cs = (ctypes.c_void_p * 1)(R.data_ptr()); su_ = (ctypes.c_void_p * 1)(sud.data_ptr())
    sv_ = (ctypes.c_void_p * 1)(svd.data_ptr()); bi_ = (ctypes.c_void_p * 1)(bd.data_ptr()); ns = (ctypes.c_int * 1)(n)
    Sout = ctypes.c_int(0); slabs = (ctypes.c_void_p * 1)()
    rc = gemv_ex(xd.data_ptr(), None, None, Bs, Cs, su_, sv_, bi_, ns, 1, m, k, K, 2, 0, ext.GEMV_OUT_ATOMIC, 3, slabs, ctypes.byref(Sout), None)
    assert rc >= 0 and Sout.value == 3
    out = torch.empty((m, n), dtype=torch.half, device=dev)
    ext.fx_finish(R, out, ss, m)
    y = o.linear_forward(x, tr, su, sv, K, 2, bias=bias, out_fp32=True).astype(np.float32)
    ref = (r0.astype(np.float32) + y)
    assert np.abs(out.float().cpu().numpy() - ref).max() / np.sqrt((y ** 2).mean()) < 1e-2
    assert np.allclose(ss.cpu().numpy(), (out.float().cpu().numpy().reshape(m, -1, 128) ** 2).sum(-1), rtol=1e-4)


@pytest.mark.parametrize("act_in", [False, True])
def test_fixed_point_residual_pipeline_carries_nan_inf_and_overflow_to_the_logits(dev, act_in):
    """the reference's fp16 residual carries NaN / Inf through to the logits (norm.cu:193-218); the fixed-point accumulator must
    not turn them into finite garbage.  (a) Inf / NaN in one input row: that row's logits are non-finite, the other row's are bit-identical to a clean
    run; (b) an o_proj whose outputs overflow the accumulator's range (svh x 3e6: |o| >= 2^20 poisons the accumulator instead of wrapping
    __double2ll_rn): non-finite logits; (c) the same model with a merely LARGE jump (x 300, representable) still matches the oracle."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    shape = LlamaShape("tiny", 512, 1024, 2, 4, 2, 128, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=2048)
    model.alloc_state(2, pos=100)
    model.fx_act_in_gemv = act_in
    clean = model.decode_step_fx().float().cpu().numpy().copy()
    assert np.isfinite(clean).all()
    x_keep = model.x0.clone()
    for bad in (float("inf"), float("nan"), -float("inf")):
        model.x0.copy_(x_keep); model.x0[0, 5] = bad
        lg = model.decode_step_fx().float().cpu().numpy()
        assert not np.isfinite(lg[0]).any(), bad                           # the Hadamard spreads the bad value over the block, the linears over every output
        assert np.array_equal(lg[1], clean[1]), bad
    model.x0.copy_(x_keep)
    sv_keep = model.layers[0]["o"].svh.clone()
    model.layers[0]["o"].svh.mul_(300.0)                                    # (c) first: large but representable
    lg = model.decode_step_fx().float().cpu().numpy().copy()
    ref = _oracle_decode(model, _np(model.x0))
    assert np.isfinite(lg).all() and np.abs(lg - ref).max() / np.sqrt((ref ** 2).mean()) < 3e-2
    model.layers[0]["o"].svh.copy_(sv_keep.float().mul(3e6).clamp(-6e4, 6e4).half())     # (b): |o| far beyond 2^20
    lg = model.decode_step_fx().float().cpu().numpy()
    assert not np.isfinite(lg).any()
    model.layers[0]["o"].svh.copy_(sv_keep)
    assert np.array_equal(model.decode_step_fx().float().cpu().numpy(), clean)          # nothing sticks: the accumulator is re-initialised every step


def test_fx_zero_request_never_outlives_the_next_launch(dev):
    """exl3_fx_zero_next is one-shot and tied to the launch that follows on this thread -- a following launch that cannot clear the
    buffer (generation 3: 16 rows) fails loudly and DROPS the request, so a later unrelated generation-4 launch leaves the buffer alone."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import _rand_linear
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    L = _rand_linear(512, 256, 4, 2, dev, gen)
    buf = torch.full((1024,), 7, dtype=torch.int64, device=dev)
    x16 = torch.randn((16, 512), device=dev, generator=gen).half(); y16 = torch.empty((16, 256), dtype=torch.half, device=dev)
    x1 = x16[:1].contiguous(); y1 = torch.empty((1, 256), dtype=torch.half, device=dev)
    ext.fx_zero_next(buf)
    with pytest.raises(RuntimeError, match="cannot clear"):
        ext.exl3_gemm(x16, L.trellis, y16, L.suh, torch.empty_like(x16), L.svh, -1, False, True, 0)
    ext.exl3_gemm(x1, L.trellis, y1, L.suh, torch.empty_like(x1), L.svh, -1, False, True, 0)       # the dropped request must not fire here
    torch.cuda.synchronize()
    assert bool((buf == 7).all())
    ext.fx_zero_next(buf)                                                    # ... and the intended use clears it exactly once
    ext.exl3_gemm(x1, L.trellis, y1, L.suh, torch.empty_like(x1), L.svh, -1, False, True, 0)
    torch.cuda.synchronize()
    assert bool((buf == 0).all())
    buf.fill_(7)
    ext.exl3_gemm(x1, L.trellis, y1, L.suh, torch.empty_like(x1), L.svh, -1, False, True, 0)
    torch.cuda.synchronize()
    assert bool((buf == 7).all())


@pytest.mark.parametrize("hd,hq,hkv", [(128, 4, 2), (64, 8, 4)])
@pytest.mark.parametrize("bsz", [1, 3, 16])
def test_per_step_rope_and_slot_tables_are_bit_identical_to_in_kernel_computation(dev, hd, hq, hkv, bsz):
    """glue_qkv with the per-step tables of qkv_prep (sin / cos / cache row computed once per decode step) against the same kernel computing
    them per layer: logits, q and every quantized KV page bit for bit, at a position that crosses into the second page."""
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    shape = LlamaShape("tiny", 512, 1024, 2, hq, hkv, hd, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=2048)
    model.alloc_state(bsz, pos=300)
    outs = []
    for use in (False, True):
        model.use_qkv_tab = use
        for c, s in model.kcache + model.vcache:
            c.zero_(); s.zero_()
        model.q.zero_()
        lg = model.decode_step_fused().clone()
        outs.append((lg, model.q.clone(), [(c.clone(), s.clone()) for c, s in model.kcache + model.vcache]))
    (l0, q0, kv0), (l1, q1, kv1) = outs
    assert torch.equal(l0, l1) and torch.equal(q0, q1)
    assert any(bool(c.any()) for c, _ in kv1)
    for (c0, s0), (c1, s1) in zip(kv0, kv1):
        assert torch.equal(c0, c1) and torch.equal(s0, s1)


@pytest.mark.parametrize("cb", [0, 2])
@pytest.mark.parametrize("bsz", [5, 16])
def test_folded_resid_rotate_pipeline_matches_oracle_and_separate_launches(dev, cb, bsz):
    """Batches above 4 rows: glue_resid_rotate (residual add + per-block sums + rotation with the PREVIOUS residual's 1/rms, consumers rescaled by
    r_new / r_prev in glue_qkv_rs / glue_act_rs) against the oracle and against the separate glue_resid + glue_rotate launches; also with a
    residual 50x smaller than the sublayer outputs (r_prev far from r_new); graph replay reproduces the eager bits."""
    from exllamav3_amd import ext
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    ext.set_gemv_variant(1)
    shape = LlamaShape("tiny", 512, 1024, 3, 4, 2, 128, 384)
    model = SyntheticEXL3Llama(shape, K=4, cb=cb, device=dev, kv_bits=4, max_ctx=2048)
    model.alloc_state(bsz, pos=700)
    for scale in (1.0, 0.02):
        model.x0.mul_(scale)
        model.fold_rotate = False
        ls = model.decode_step_fused().float().cpu().numpy().copy()
        xs = model.x.float().cpu().numpy().copy()
        model.fold_rotate = True
        lf = model.decode_step_fused().float().cpu().numpy().copy()
        assert np.isfinite(lf).all()
        ref = _oracle_decode(model, _np(model.x0))
        rms = np.sqrt((ref ** 2).mean())
        assert np.abs(lf - ref).max() / rms < 3e-2
        assert np.abs(lf - ls).max() / rms < 1.5e-2
        xf = model.x.float().cpu().numpy()
        assert np.abs(xf - xs).max() / np.sqrt((xs ** 2).mean()) < 1e-2
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            model.decode_step_fused()
    for _ in range(3):
        model.logits.zero_(); g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(model.logits.float().cpu().numpy(), lf)


@pytest.mark.parametrize("bsz", [1, 3])
def test_mixtral_fused_moe_tail_and_router_norm_agree_with_separate_launches(dev, bsz):
    """The one-rank MoE block in 4 launches (router with the RMSNorm inside, gate|up, deferred weighted down, glue_resid_moe) against the same
    block with separate rms_norm / split-k reduce / slot sum / residual launches: logits and residual stream agree to fp16 rounding (the
    deferred launch picks another k-split, so the fp32 sums are ordered differently); graph replay reproduces the eager bits."""
    from exllamav3_amd.mixtral_path import MixtralShape, SyntheticEXL3Mixtral
    shape = MixtralShape("tiny-moe", 512, 768, 3, 4, 2, 128, 384, 8, 2)
    model = SyntheticEXL3Mixtral(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=1024)
    model.alloc_state(bsz, pos=77)
    model.norm_in_router, model.fused_moe_tail = False, False
    l0 = model.decode_step().float().cpu().numpy().copy(); x0 = model.x.float().cpu().numpy().copy()
    model.norm_in_router, model.fused_moe_tail = True, True
    l1 = model.decode_step().float().cpu().numpy().copy(); x1 = model.x.float().cpu().numpy().copy()
    rms = np.sqrt((l0 ** 2).mean())
    assert np.isfinite(l1).all() and np.abs(l1 - l0).max() / rms < 1e-2
    assert np.abs(x1 - x0).max() / np.sqrt((x0 ** 2).mean()) < 1e-2
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            model.decode_step()
    for _ in range(2):
        model.logits.zero_(); g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(model.logits.float().cpu().numpy(), l1)


@pytest.mark.parametrize("cb", [0, 2])
@pytest.mark.parametrize("with_attention", [False, True])
def test_mixtral_fixed_point_pipeline_matches_the_launch_per_op_moe_block(dev, cb, with_attention):
    """mixtral_path.decode_step_fx (batch 1: 6 launches per layer, the MoE block in 3 -- router on the fixed-point residual, indexed gate|up launch leaving
    slabs, indexed down launch adding its weighted rows into the accumulator with integer atomics) against decode_step with every stage as its own
    launch: logits and final residual agree to fp16 rounding; the same experts are routed in every layer; graph replay and a second eager run
    reproduce the bits (integer atomics: no dependence on arrival order)."""
    from exllamav3_amd.mixtral_path import MixtralShape, SyntheticEXL3Mixtral
    shape = MixtralShape("tiny-moe", 512, 896, 3, 4, 2, 128, 384, 8, 2)      # inter / 128 = 7 blocks: odd unit counts in the down launch
    model = SyntheticEXL3Mixtral(shape, K=4, cb=cb, device=dev, kv_bits=4, max_ctx=1024)
    model.alloc_state(1, pos=77)
    model.with_attention = with_attention
    model.norm_in_router, model.fused_moe_tail = False, False
    l0 = model.decode_step().float().cpu().numpy().copy(); x0 = model.x.float().cpu().numpy().copy()
    sel0 = [L["moe"].sel.cpu().numpy().copy() for L in model.layers]
    l1 = model.decode_step_fx().float().cpu().numpy().copy(); x1 = model.x.float().cpu().numpy().copy()
    sel1 = [L["moe"].sel.cpu().numpy().copy() for L in model.layers]
    assert all(np.array_equal(a, b) for a, b in zip(sel0, sel1))
    rms = np.sqrt((l0 ** 2).mean())
    assert np.isfinite(l1).all() and np.abs(l1 - l0).max() / rms < 1e-2, np.abs(l1 - l0).max() / rms
    assert np.abs(x1 - x0).max() / np.sqrt((x0 ** 2).mean()) < 1e-2
    l2 = model.decode_step_fx().float().cpu().numpy().copy()
    assert np.array_equal(l2, l1)
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            model.decode_step_fx()
    for _ in range(2):
        model.logits.zero_(); g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(model.logits.float().cpu().numpy(), l1)


@pytest.mark.parametrize("bsz", [1, 3])
@pytest.mark.parametrize("with_attention", [False, True])
def test_mixtral_layer_path_matches_oracle(dev, bsz, with_attention):
    """mixtral_path.SyntheticEXL3Mixtral (config 5's layer: attention sublayer + sparse-MoE sublayer with top-2 routing, 4-bit KV) against the
    oracle composition, with and without the decode attention over the quantized cache; graph replay reproduces the eager bits."""
    from exllamav3_amd.mixtral_path import MixtralShape, SyntheticEXL3Mixtral
    shape = MixtralShape("tiny-moe", 256, 384, 2, 4, 2, 128, 384, 6, 2)
    model = SyntheticEXL3Mixtral(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=1024)
    pos = 300
    model.alloc_state(bsz, pos=pos)
    model.with_attention = with_attention
    rng = np.random.default_rng(9)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ctx = []
    if with_attention:
        for li in range(shape.layers):      # random context in every layer's cache (contiguous pages per sequence in this model)
            ck = rng.standard_normal((bsz, 1024, model.hkv * 128)).astype(np.float16); cv = rng.standard_normal((bsz, 1024, model.hkv * 128)).astype(np.float16)
            kq, ks = o.kv_quant(ck, 4); vq, vs = o.kv_quant(cv, 4)
            kc, ksc = model.kcache[li]; vc, vsc = model.vcache[li]
            kc.copy_(T(kq.view(np.int32)).view(kc.shape)); ksc.copy_(T(ks).view(ksc.shape)); vc.copy_(T(vq.view(np.int32)).view(vc.shape)); vsc.copy_(T(vs).view(vsc.shape))
            ctx.append((kq, ks, vq, vs))
    logits = model.decode_step().float().cpu().numpy().copy()
    assert np.isfinite(logits).all()
    # ---- oracle
    x = _np(model.x0)
    pend = None
    for li, L in enumerate(model.layers):
        if pend is None: xn = o.rms_norm(x, _np(L["norm1"]), model.eps)
        else: xn, x = o.rms_norm(pend, _np(L["norm1"]), model.eps, residual_in=x)
        q, k, v = _lin(L["q"], xn), _lin(L["k"], xn), _lin(L["v"], xn)
        q4, k4 = o.rope(q.reshape(bsz, 1, model.hq, 128), k.reshape(bsz, 1, model.hkv, 128), _np(model.inv_freq), positions=_np(model.positions), rope_mode=o.ROPE_NEOX)
        att = q4.reshape(bsz, -1)
        if with_attention:
            kq, ks, vq, vs = ctx[li]
            knq, kns = o.kv_quant(k4.reshape(bsz, 1, -1), 4); vnq, vns = o.kv_quant(v.reshape(bsz, 1, -1), 4)
            kq[:, pos:pos + 1] = knq; ks[:, pos:pos + 1] = kns; vq[:, pos:pos + 1] = vnq; vs[:, pos:pos + 1] = vns
            kd = o.kv_dequant(kq, ks, 4).reshape(bsz, 1024, model.hkv, 128); vd = o.kv_dequant(vq, vs, 4).reshape(bsz, 1024, model.hkv, 128)
            att = o.attn_decode_qcache(q4.reshape(bsz, model.hq, 128), kd, vd, [pos + 1] * bsz).reshape(bsz, -1)
        ov = _lin(L["o"], att, out_fp32=True)
        xn, x = o.rms_norm(ov, _np(L["norm2"]), model.eps, residual_in=x)
        moe = L["moe"]
        _, sel, w = o.routing_std(xn, _np(moe.router), 2)
        y = np.zeros((bsz, 256), dtype=np.float32)
        for t in range(bsz):
            for j in range(2):
                e = int(sel[t, j])
                g = _lin(moe.gate[e], xn[t:t + 1]).astype(np.float32); u = _lin(moe.up[e], xn[t:t + 1]).astype(np.float32)
                a = (g / (1 + np.exp(-g)) * u).astype(np.float16)
                y[t] += float(w[t, j]) * _lin(moe.down[e], a, out_fp32=True)[0]
        pend = y
    xn, x = o.rms_norm(pend, _np(model.final_norm), model.eps, residual_in=x)
    ref = _lin(model.lm_head, xn).astype(np.float32)
    assert np.abs(logits - ref).max() / np.sqrt((ref ** 2).mean()) < 3e-2
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_, stream=st):
            model.decode_step()
    model.logits.zero_(); g_.replay(); torch.cuda.synchronize()
    assert np.array_equal(model.logits.float().cpu().numpy(), logits)


@pytest.mark.parametrize("grouped", [True, False])
@pytest.mark.parametrize("tokens,experts,top_k", [(600, 6, 2), (1200, 8, 2), (160, 4, 3)])
def test_moe_prefill_block_matches_oracle(dev, tokens, experts, top_k, grouped):
    """SyntheticEXL3MoE.forward_prefill (round 4; the grouped-by-expert large-batch tier of modules/block_sparse_mlp.py:1169-1330: router over all rows,
    per expert gather -> gate|up GEMM + silu * mul -> down GEMM, fixed-order weighted scatter) against the oracle per token: routing identical, outputs
    to 2e-2 of their RMS; experts land on both sides of the 144-row switch between the small-m kernels and the reconstruct + GEMM route; a second
    run reproduces the bits (no atomic scatter).  grouped: the block as two grouped GEMMs over all experts (ext.gemm_nt_grouped, row boundaries on the device, silu * mul
    in the first one's epilogue, fp32 output from the second) -- no host synchronisation inside; not grouped: the per-expert loop."""
    from exllamav3_amd.moe_path import SyntheticEXL3MoE
    moe = SyntheticEXL3MoE(256, 384, experts=experts, top_k=top_k, K=4, cb=2, device=dev, seed=21)
    moe.grouped_prefill = grouped
    x = torch.randn((tokens, 256), device=dev, generator=torch.Generator(device=dev).manual_seed(tokens)).half()
    y = moe.forward_prefill(x)
    y1 = y.clone()
    assert torch.equal(moe.forward_prefill(x), y1)
    xs = _np(x)
    _, sel, w = o.routing_std(xs, _np(moe.router), top_k)
    assert np.array_equal(_np(moe.pf_sel), sel)
    counts = np.bincount(sel.reshape(-1), minlength=experts)
    if tokens >= 600: assert counts.max() > 144                    # the reconstruct + GEMM route is exercised
    ref = np.zeros((tokens, 256), dtype=np.float32)
    for e in range(experts):
        tok, slot = np.nonzero(sel == e)
        if not len(tok): continue
        g = _lin(moe.gate[e], xs[tok]).astype(np.float32); u = _lin(moe.up[e], xs[tok]).astype(np.float32)
        a = (g / (1 + np.exp(-g)) * u).astype(np.float16)
        d = _lin(moe.down[e], a, out_fp32=True)
        ref[tok] += w[tok, slot].astype(np.float32)[:, None] * d
    assert np.isfinite(_np(y1)).all()
    assert np.abs(_np(y1) - ref).max() / np.sqrt((ref ** 2).mean()) < 2e-2


def test_moe_prefill_grouped_block_is_capturable(dev):
    """The grouped branch of forward_prefill has no host synchronisation (row boundaries stay on the device): it records into a hipGraph, and a replay over OTHER activations
    (other routing, other per-expert row counts) gives what the eager call gives."""
    from exllamav3_amd.moe_path import SyntheticEXL3MoE
    moe = SyntheticEXL3MoE(256, 384, experts=8, top_k=2, K=4, cb=2, device=dev, seed=9)
    gen = torch.Generator(device=dev); gen.manual_seed(3)
    x = torch.randn((700, 256), device=dev, generator=gen).half()
    moe.forward_prefill(x); torch.cuda.synchronize()                      # (first-call set-up of the launchers happens outside the capture)
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            y = moe.forward_prefill(x)
    x2 = torch.randn((700, 256), device=dev, generator=gen).half()
    x.copy_(x2); g.replay(); torch.cuda.synchronize()
    got = y.clone()
    want = moe.forward_prefill(x2)
    assert torch.equal(got, want)


def test_moe_prefill_expert_parallel_partials_sum_to_the_whole(dev):
    """forward_prefill under expert parallelism: the partial sums of the two halves of the experts add up to the one-rank result (same routing on every rank)."""
    from exllamav3_amd.moe_path import SyntheticEXL3MoE
    full = SyntheticEXL3MoE(256, 384, experts=8, top_k=2, K=4, cb=2, device=dev, seed=5)
    x = torch.randn((400, 256), device=dev, generator=torch.Generator(device=dev).manual_seed(1)).half()
    yg = full.forward_prefill(x).clone()                   # (every expert local: the two grouped GEMMs)
    full.grouped_prefill = False                           # the per-expert loop, the structure the ranks below run: same kernels per expert -> the sums agree to fp32 rounding
    y = full.forward_prefill(x).clone()
    assert np.abs(_np(yg) - _np(y)).max() / np.sqrt((_np(y) ** 2).mean()) < 2e-2          # grouped against the loop (the small-m kernels below 144 rows per expert there): the bar both meet against the oracle
    parts = []
    for first, last in ((0, 4), (4, 8)):
        half = SyntheticEXL3MoE(256, 384, experts=8, top_k=2, K=4, cb=2, device=dev, seed=5)
        half.gate, half.up, half.down = (l[first:last] for l in (half.gate, half.up, half.down))
        half.first, half.last = first, last
        parts.append(half.forward_prefill(x).clone())
    assert np.abs(_np(parts[0] + parts[1]) - _np(y)).max() / np.sqrt((_np(y) ** 2).mean()) < 1e-5


def test_mixtral_prefill_chunk_matches_oracle_on_sampled_rows(dev):
    """mixtral_path.prefill_chunk (attention sublayer + sparse-MoE sublayer per layer) on a small Mixtral-shaped model: residual stream after the layers on
    sampled token rows and the last token's logits against the oracle (every row is independent with the attention core out of scope)."""
    from exllamav3_amd.mixtral_path import MixtralShape, SyntheticEXL3Mixtral
    shape = MixtralShape("tiny-moe", 256, 384, 2, 4, 2, 128, 384, 6, 2)
    model = SyntheticEXL3Mixtral(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=1024)
    toks = 700
    logits = _np(model.prefill_chunk(toks).float())
    rows = np.array([0, 1, 255, 256, 511, toks - 1])
    x = _np(model.px0)[rows]
    for L in model.layers:
        xn = o.rms_norm(x, _np(L["norm1"]), model.eps)
        q, k = _lin(L["q"], xn), _lin(L["k"], xn)
        q4 = np.empty((len(rows), model.hq, 128), np.float16)
        for i, r in enumerate(rows):
            qi, _ = o.rope(q[i].reshape(1, 1, model.hq, 128), k[i].reshape(1, 1, model.hkv, 128), _np(model.inv_freq), position=int(r), rope_mode=o.ROPE_NEOX)
            q4[i] = qi[0, 0]
        ov = _lin(L["o"], q4.reshape(len(rows), -1), out_fp32=True)
        xn, x = o.rms_norm(ov, _np(L["norm2"]), model.eps, residual_in=x)
        moe = L["moe"]
        _, sel, w = o.routing_std(xn, _np(moe.router), 2)
        y = np.zeros((len(rows), 256), dtype=np.float32)
        for t in range(len(rows)):
            for j in range(2):
                e = int(sel[t, j])
                g = _lin(moe.gate[e], xn[t:t + 1]).astype(np.float32); u = _lin(moe.up[e], xn[t:t + 1]).astype(np.float32)
                a = (g / (1 + np.exp(-g)) * u).astype(np.float16)
                y[t] += float(w[t, j]) * _lin(moe.down[e], a, out_fp32=True)[0]
        x = (x.astype(np.float32) + y).astype(np.float16)
    got_x = _np(model.px_out)[rows].astype(np.float32)
    assert np.abs(got_x - x.astype(np.float32)).max() / np.sqrt((x.astype(np.float32) ** 2).mean()) < 2e-2
    xl = o.rms_norm(x[-1:], _np(model.final_norm), model.eps)
    ref = _lin(model.lm_head, xl).astype(np.float32)
    assert np.isfinite(logits).all() and np.abs(logits - ref).max() / np.sqrt((ref ** 2).mean()) < 5e-2
