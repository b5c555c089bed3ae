"""GPU parity of the prefill-side ops: reconstruct_had_slice, hgemm, and the LinearEXL3 host paths."""
import numpy as np
import pytest
import torch
from oracle import exl3_oracle as o

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("k,n,K,cb", [(256, 128, 3, 0), (512, 384, 2, 0), (1024, 512, 5, 0), (384, 256, 4, 1), (256, 512, 3, 2),
                                      (4096, 1024, 3, 0), (128, 128, 1, 2), (256, 256, 8, 0), (128, 256, 6, 1), (256, 128, 7, 2)])
def test_reconstruct_had_slice(dev, k, n, K, cb):
    """tests/test_reconstruct_had.py:35-68 (shapes, seeds k*7+n+K, +-1 scales) + extra bitrates; max-abs/max < 2e-3."""
    from exllamav3_amd import ext
    tr, suh, svh = o.synth_linear(k, n, K)
    ref = o.weight_tensor(tr, suh, svh, K, cb)
    w = torch.empty((k, n), dtype=torch.half, device=dev)
    ext.reconstruct_had_slice(w, _t(tr, dev), _t(suh, dev), _t(svh, dev), K, cb == 1, cb == 2, 0)
    got = w.float().cpu().numpy()
    assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-3
    if n >= 256:    # slice offset path (test_reconstruct_had.py:54-67)
        ws = torch.empty((k, 128), dtype=torch.half, device=dev)
        ext.reconstruct_had_slice(ws, _t(tr, dev), _t(suh, dev), _t(svh[128:], dev), K, cb == 1, cb == 2, 128)
        assert np.abs(ws.float().cpu().numpy() - ref[:, 128:256]).max() / np.abs(ref).max() < 2e-3


def test_reconstruct_had_real_valued_scales(dev):
    from exllamav3_amd import ext
    k, n, K, cb = 512, 256, 4, 2
    tr, suh, svh = o.synth_linear(k, n, K, realistic=True)
    ref = o.weight_tensor(tr, suh, svh, K, cb)
    w = torch.empty((k, n), dtype=torch.half, device=dev)
    ext.reconstruct_had_slice(w, _t(tr, dev), _t(suh, dev), _t(svh, dev), K, False, True, 0)
    assert np.abs(w.float().cpu().numpy() - ref).max() / np.abs(ref).max() < 2e-3


@pytest.mark.parametrize("m,k,n", [(1, 128, 128), (17, 256, 384), (256, 1024, 512), (1000, 512, 640)])
def test_hgemm(dev, m, k, n):
    from exllamav3_amd import ext
    rng = np.random.default_rng(m + k)
    a = rng.standard_normal((m, k)).astype(np.float16); b = (rng.standard_normal((k, n)) * 0.1).astype(np.float16)
    ref = a.astype(np.float32) @ b.astype(np.float32)
    for dt in (torch.half, torch.float):
        c = torch.full((m, n), float("nan"), dtype=dt, device=dev)
        ext.hgemm(_t(a, dev), _t(b, dev), c)
        assert np.allclose(c.float().cpu().numpy(), ref, rtol=2e-2, atol=2e-2)
    # column-slice output (hgemm.cu: c.stride(-2))
    cbig = torch.zeros((m, n + 256), dtype=torch.half, device=dev)
    ext.hgemm(_t(a, dev), _t(b, dev), cbig[:, 128:128 + n])
    assert np.allclose(cbig[:, 128:128 + n].float().cpu().numpy(), ref, rtol=2e-2, atol=2e-2)
    assert float(cbig[:, :128].abs().max()) == 0.0 and float(cbig[:, 128 + n:].abs().max()) == 0.0


@pytest.mark.parametrize("rows", [1, 16, 144, 145, 512, 1024, 1500])
def test_linear_exl3_paths_agree(dev, rows):
    """modules/quant/exl3.py:114-218: kernel path (rows <= 144), unfused reconstruct path and fused-W path against the oracle, with this
    library's switch point (fused from 145 rows) and the reference's (1024 rows); tests/test_qgemm.py tolerance rtol = atol = 0.05."""
    from exllamav3_amd import linear
    from exllamav3_amd.linear import LinearEXL3
    k, n, K, cb = 512, 384, 4, 2
    tr, suh, svh = o.synth_linear(k, n, K, realistic=True)
    bias = (np.random.default_rng(1).standard_normal(n) * 0.1).astype(np.float16)
    lin = LinearEXL3(k, n, _t(tr, dev), _t(suh, dev), _t(svh, dev), mul1=True, bias=_t(bias, dev))
    x = np.random.default_rng(0).standard_normal((rows, k)).astype(np.float16)
    ref = o.linear_forward(x, tr, suh, svh, K, cb, bias=bias).astype(np.float32)
    y = lin.forward(_t(x, dev), {}).float().cpu().numpy()
    assert np.allclose(y, ref, rtol=0.05, atol=0.05)
    err = np.abs(y - ref).max() / np.sqrt((ref ** 2).mean())
    assert err < 2e-2
    y2 = lin.forward(_t(x, dev), {"reconstruct": True}).float().cpu().numpy()
    assert np.allclose(y2, ref, rtol=0.05, atol=0.05)
    old = linear.FUSED_RECONSTRUCT_MIN_ROWS
    linear.FUSED_RECONSTRUCT_MIN_ROWS = 1024                     # exl3.py:184
    try:
        y3 = lin.forward(_t(x, dev), {"reconstruct": True}).float().cpu().numpy()
    finally:
        linear.FUSED_RECONSTRUCT_MIN_ROWS = old
    assert np.allclose(y3, ref, rtol=0.05, atol=0.05)


def test_linear_exl3_lm_head_slicing(dev):
    """out_features > MAX_RECONSTRUCT_SLICE_N is processed in column slices (exl3.py:199-211)."""
    from exllamav3_amd import linear
    k, n, K, cb = 256, 1024, 3, 0
    tr, suh, svh = o.synth_linear(k, n, K)
    lin = linear.LinearEXL3(k, n, _t(tr, dev), _t(suh, dev), _t(svh, dev))
    x = np.random.default_rng(0).standard_normal((1100, k)).astype(np.float16)
    ref = o.linear_forward(x, tr, suh, svh, K, cb).astype(np.float32)
    old = linear.MAX_RECONSTRUCT_SLICE_N
    linear.MAX_RECONSTRUCT_SLICE_N = 384
    try:
        for rows in (200, 1100):
            y = lin.forward(_t(x[:rows], dev), {}).float().cpu().numpy()
            assert np.allclose(y, ref[:rows], rtol=0.05, atol=0.05)
        old_f = linear.FUSED_RECONSTRUCT_MIN_ROWS
        linear.FUSED_RECONSTRUCT_MIN_ROWS = 1024                 # the unfused slices (reference switch point)
        try:
            y = lin.forward(_t(x[:200], dev), {}).float().cpu().numpy()
        finally:
            linear.FUSED_RECONSTRUCT_MIN_ROWS = old_f
        assert np.allclose(y, ref[:200], rtol=0.05, atol=0.05)
    finally:
        linear.MAX_RECONSTRUCT_SLICE_N = old


@pytest.mark.parametrize("m,k,n", [(1024, 512, 384), (2048, 256, 1024)])
def test_hgemm_acc_and_forward_add_residual(dev, m, k, n):
    """hgemm_acc: c = fp16(a @ b + c) equals fp32-output hgemm followed by the fp16 residual add (same single rounding up to the
    GEMM's fp32 summation order); LinearEXL3.forward_add_residual against forward(out fp32) + add, and the oracle."""
    from exllamav3_amd import ext
    from exllamav3_amd.linear import LinearEXL3
    rng = np.random.default_rng(m + n)
    a = _t(rng.standard_normal((m, k)).astype(np.float16), dev)
    b = _t((rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float16), dev)
    r0 = _t(rng.standard_normal((m, n)).astype(np.float16), dev)
    y = torch.empty((m, n), dtype=torch.float, device=dev)
    ext.hgemm(a, b, y)
    ref = (r0.float() + y).half()
    r1 = r0.clone()
    ext.hgemm_acc(a, b, r1)
    d = (r1.float() - ref.float()).abs()
    assert float(d.max()) <= 2e-3 * float(ref.float().abs().max())          # at most an fp16 ulp where the fp32 sums differ in order
    assert float((d > 0).float().mean()) < 0.02
    tr, suh, svh = o.synth_linear(k, n, 4, realistic=True)
    lin = LinearEXL3(k, n, _t(tr, dev), _t(suh, dev), _t(svh, dev), mul1=True)
    x = _t(rng.standard_normal((m, k)).astype(np.float16), dev)
    ra = r0.clone(); rb = r0.clone()
    lin.forward_add_residual(x, ra)
    ext.add(rb, lin.forward(x, out_dtype=torch.float))
    assert float((ra.float() - rb.float()).abs().max()) <= 4e-3 * float(rb.float().abs().max())
    ref2 = (r0.float().cpu().numpy() + o.linear_forward(x.cpu().numpy(), tr, suh, svh, 4, 2, out_fp32=True)).astype(np.float16).astype(np.float32)
    assert np.abs(ra.float().cpu().numpy() - ref2).max() / np.sqrt((ref2 ** 2).mean()) < 2e-2


@pytest.mark.parametrize("K,cb", [(4, 2), (3, 0), (5, 1)])
def test_reconstruct_had_multi_matches_separate_launches(dev, K, cb):
    """reconstruct_had_multi_t (q | k | v of the fused prefill GEMM in ONE launch) is bit-identical to one reconstruct_had_slice_t per matrix into the
    same stacked buffer; matrices of different width, 1 to 4 of them; a row stride wider than k."""
    from exllamav3_amd import ext
    k = 512
    mats = [o.synth_linear(k, n, K, seed=40 + i, realistic=True) for i, n in enumerate((384, 128, 128, 256))]
    dm = [tuple(_t(a, dev) for a in t) for t in mats]
    for cnt in (1, 3, 4):
        ntot = sum(t[0].shape[1] * 16 for t in mats[:cnt])
        sep = torch.zeros((ntot, k + 64), dtype=torch.half, device=dev); one = torch.zeros_like(sep)
        n0 = 0
        for tr, su, sv in dm[:cnt]:
            n = tr.shape[1] * 16
            ext.reconstruct_had_slice_t(sep[n0: n0 + n, :k], tr, su, sv, K, cb == 1, cb == 2, 0)
            n0 += n
        ext.reconstruct_had_multi_t(one[:, :k], [t[0] for t in dm[:cnt]], [t[1] for t in dm[:cnt]], [t[2] for t in dm[:cnt]], K, cb == 1, cb == 2)
        assert torch.equal(sep, one), cnt
    ref = o.weight_tensor(*mats[1], K, cb)
    got = one[384:512, :k].t().float().cpu().numpy()
    assert np.abs(got - ref).max() / np.abs(ref).max() < 2e-3


@pytest.mark.parametrize("k,n,K,cb", [(256, 384, 4, 2), (512, 256, 3, 0), (1024, 128, 6, 1)])
def test_reconstruct_had_slice_transposed(dev, k, n, K, cb):
    """reconstruct_had_slice_t writes W^T (row stride >= k): bit-identical to the transpose of reconstruct_had_slice, whole matrix, a column
    slice, and into a wider row-strided buffer shared by two matrices."""
    from exllamav3_amd import ext
    tr, suh, svh = o.synth_linear(k, n, K, realistic=True)
    ttr, tsu, tsv = _t(tr, dev), _t(suh, dev), _t(svh, dev)
    w = torch.empty((k, n), dtype=torch.half, device=dev)
    ext.reconstruct_had_slice(w, ttr, tsu, tsv, K, cb == 1, cb == 2, 0)
    wt = torch.empty((n, k), dtype=torch.half, device=dev)
    ext.reconstruct_had_slice_t(wt, ttr, tsu, tsv, K, cb == 1, cb == 2, 0)
    ref = o.weight_tensor(tr, suh, svh, K, cb)
    # same arithmetic up to the compiler's choice of single- vs double-rounded float -> half conversions: at most one fp16 ulp apart
    ulp = 2.0 ** -10 * float(np.abs(ref).max())
    assert float((wt.float() - w.t().float()).abs().max()) <= ulp
    assert np.abs(wt.t().float().cpu().numpy() - ref).max() / np.abs(ref).max() < 2e-3
    if n >= 256:
        ws = torch.empty((128, k), dtype=torch.half, device=dev)
        ext.reconstruct_had_slice_t(ws, ttr, tsu, tsv[128:], K, cb == 1, cb == 2, 128)
        assert float((ws.float() - w[:, 128:256].t().float()).abs().max()) <= ulp
    wide = torch.zeros((n, k + 64), dtype=torch.half, device=dev)
    ext.reconstruct_had_slice_t(wide[:, :k], ttr, tsu, tsv, K, cb == 1, cb == 2, 0)
    assert torch.equal(wide[:, :k], wt) and float(wide[:, k:].abs().max()) == 0.0


@pytest.mark.parametrize("m,k,n", [(1024, 512, 384), (2048, 1024, 256)])
def test_hgemm_nt(dev, m, k, n):
    from exllamav3_amd import ext
    rng = np.random.default_rng(m + k + n)
    a = _t(rng.standard_normal((m, k)).astype(np.float16), dev)
    b = _t((rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float16), dev)
    bt = b.t().contiguous()
    ref = a.float() @ b.float()
    for route in (None, "library", "own"):                # default = the own kernel where its tiles apply (fp16 output, rows >= 256), the library otherwise
        for dt in (torch.half, torch.float):
            c = torch.empty((m, n), dtype=dt, device=dev)
            ext.hgemm_nt(a, bt, c, route=route)
            assert float((c.float() - ref).abs().max()) < 2e-2
        r0 = _t(rng.standard_normal((m, n)).astype(np.float16), dev)
        r1 = r0.clone()
        ext.hgemm_nt(a, bt, r1, accumulate=True, route=route)
        assert float((r1.float() - (r0.float() + ref)).abs().max()) < 3e-2


def test_reconstruct_had_multi_t_interleaved_is_a_block_permutation(dev):
    """exl3_reconstruct_had_multi_t_interleaved: the same values as the stacked form, the matrices' 128-row blocks alternating (bit-exact)."""
    from exllamav3_amd import ext
    k, n, K = 256, 384, 4
    mats = [o.synth_linear(k, n, K, seed=s_, realistic=True) for s_ in (3, 4)]
    tr = [_t(m[0], dev) for m in mats]; su = [_t(m[1], dev) for m in mats]; sv = [_t(m[2], dev) for m in mats]
    st = torch.empty((2 * n, k), dtype=torch.half, device=dev); il = torch.full((2 * n, k), float("nan"), dtype=torch.half, device=dev)
    ext.reconstruct_had_multi_t(st, tr, su, sv, K, False, True)
    ext.reconstruct_had_multi_t(il, tr, su, sv, K, False, True, True)
    want = st.view(2, n // 128, 128, k).permute(1, 0, 2, 3).reshape(2 * n, k)
    assert torch.equal(il, want)
    with pytest.raises(RuntimeError):
        ext.reconstruct_had_multi_t(il[: n + 128], [tr[0], _t(o.synth_linear(k, 128, K, seed=5)[0], dev)], su, sv, K, False, True, True)      # different n


@pytest.mark.parametrize("rows", [1024, 200])
def test_forward_gate_up_silu_fused_gemm(dev, rows):
    """gate|up + silu·mul of the prefill route against two forwards + silu_mul and the oracle: from 256 rows ONE own GEMM whose epilogue applies silu(g) * u to the tile
    (W^T with gate / up blocks interleaved); below, one library GEMM + silu_mul_2d."""
    from exllamav3_amd import ext
    from exllamav3_amd.linear import LinearEXL3
    k, n, K = 256, 384, 4
    g = o.synth_linear(k, n, K, seed=1, realistic=True); u = o.synth_linear(k, n, K, seed=2, realistic=True)
    lg = LinearEXL3(k, n, _t(g[0], dev), _t(g[1], dev), _t(g[2], dev), mul1=True)
    lu = LinearEXL3(k, n, _t(u[0], dev), _t(u[1], dev), _t(u[2], dev), mul1=True)
    x = np.random.default_rng(0).standard_normal((rows, k)).astype(np.float16)
    a = LinearEXL3.forward_gate_up_silu(lg, lu, _t(x, dev))
    a2 = torch.empty_like(a)
    ext.silu_mul(lg.forward(_t(x, dev)), lu.forward(_t(x, dev)), a2)
    assert float((a.float() - a2.float()).abs().max()) <= 4e-3 * float(a2.float().abs().max()) + 1e-3
    gf = o.linear_forward(x, g[0], g[1], g[2], K, 2).astype(np.float32); uf = o.linear_forward(x, u[0], u[1], u[2], K, 2).astype(np.float32)
    ref = (gf / (1 + np.exp(-gf)) * uf)
    assert np.abs(a.float().cpu().numpy() - ref).max() / np.sqrt((ref ** 2).mean()) < 3e-2


def test_loaded_checkpoint_linear_runs_on_gpu(dev, tmp_path):
    """safetensors file -> loader.load_linear_exl3 (tensors land on the GPU as stored) -> forward, against the oracle."""
    from safetensors.torch import save_file
    from exllamav3_amd import loader
    k, n, K = 256, 384, 4
    tr, suh, svh = o.synth_linear(k, n, K, seed=8, realistic=True)
    save_file({"m.q_proj.trellis": torch.from_numpy(tr), "m.q_proj.suh": torch.from_numpy(suh), "m.q_proj.svh": torch.from_numpy(svh),
               "m.q_proj.mcg": torch.zeros(1, dtype=torch.int32)}, str(tmp_path / "model.safetensors"))
    lin = loader.load_linear_exl3(loader.SafetensorsCollection(str(tmp_path)), "m.q_proj", dev)
    assert lin.mcg and not lin.mul1 and lin.trellis.device.type == "cuda"
    x = np.random.default_rng(0).standard_normal((3, k)).astype(np.float16)
    y = lin.forward(_t(x, dev)).float().cpu().numpy()
    ref = o.linear_forward(x, tr, suh, svh, K, 1).astype(np.float32)
    assert np.abs(y - ref).max() / np.sqrt((ref ** 2).mean()) < 1e-2


def test_fused_qkv_route_strided_consumers(dev):
    """LinearEXL3.forward_multi (one GEMM for q|k|v) and its strided consumers against the separate-forward route: column ranges equal the
    separate GEMMs up to fp32 summation order, rope_strided == rope and quant_cache_paged_strided == quant_cache_paged bit for bit on the
    same values, hgemm_nt with a strided A operand == the contiguous call."""
    from exllamav3_amd import ext
    from exllamav3_amd.linear import LinearEXL3
    tokens, h, hq, hkv, hd = 300, 512, 4, 2, 128
    rng = np.random.default_rng(3)
    lins = []
    for n, K, cb in ((hq * hd, 4, 2), (hkv * hd, 3, 0), (hkv * hd, 5, 2)):               # per-matrix bitrates / codebooks are free on this route
        tr, suh, svh = o.synth_linear(h, n, K, seed=n + K, realistic=True)
        lins.append(LinearEXL3(h, n, _t(tr, dev), _t(suh, dev), _t(svh, dev), mcg=(cb == 1), mul1=(cb == 2)))
    x = _t(rng.standard_normal((tokens, h)).astype(np.float16), dev)
    outs = LinearEXL3.forward_multi(lins, x)
    assert outs is not None and [tuple(t.shape) for t in outs] == [(tokens, 512), (tokens, 256), (tokens, 256)] and outs[0].stride(0) == 1024
    sep = [l.forward(x) for l in lins]
    for a, b in zip(outs, sep):
        assert float((a.float() - b.float()).abs().max()) < 2e-2 * float(b.float().abs().max()) + 1e-3
    assert LinearEXL3.forward_multi(lins, x[:100]) is None                              # below the reconstruct threshold: caller falls back
    # rope in place on the column ranges == rope on contiguous copies
    inv = (1.0 / (500000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))).to(dev)
    qc, kc = outs[0].contiguous().view(1, tokens, hq, hd), outs[1].contiguous().view(1, tokens, hkv, hd)
    ext.rope(qc, qc, kc, kc, inv, 5, None, None, 2, 1.0)
    ext.rope_strided(outs[0], outs[1], inv, 5, None, None, 1.0, 1, tokens)
    assert torch.equal(outs[0], qc.view(tokens, -1)) and torch.equal(outs[1], kc.view(tokens, -1))
    # paged KV append from the column ranges == from contiguous copies
    G, pages, bits = hkv * hd // 32, 2, 4
    bt = torch.arange(pages, dtype=torch.int32, device=dev).view(1, pages); sl = torch.zeros((1,), dtype=torch.int32, device=dev)
    mk = lambda: [torch.zeros((pages, 256, G * bits), dtype=torch.int32, device=dev) for _ in range(2)] + [torch.zeros((pages, 256, G), dtype=torch.half, device=dev) for _ in range(2)]
    c0, c1 = mk(), mk()
    ext.quant_cache_paged(outs[1].contiguous().view(1, tokens, -1), c0[0], c0[2], outs[2].contiguous().view(1, tokens, -1), c0[1], c0[3], sl, bt, 256, tokens)
    ext.quant_cache_paged_strided(outs[1], c1[0], c1[2], outs[2], c1[1], c1[3], sl, bt, 256, tokens)
    for a, b in zip(c0, c1):
        assert torch.equal(a, b)
    # o_proj from the strided q columns
    tr, suh, svh = o.synth_linear(hq * hd, h, 4, seed=9, realistic=True)
    lo = LinearEXL3(hq * hd, h, _t(tr, dev), _t(suh, dev), _t(svh, dev), mul1=True)
    r0 = _t(rng.standard_normal((tokens, h)).astype(np.float16), dev); r1 = r0.clone()
    lo.forward_add_residual(outs[0], r0)
    lo.forward_add_residual(outs[0].contiguous(), r1)
    assert float((r0.float() - r1.float()).abs().max()) < 2e-2 * float(r1.float().abs().max())


@pytest.mark.parametrize("dt", [torch.half, torch.float])
def test_softcap_matches_oracle(dev, dt):
    """ext.softcap (softcap.cu): y = scale * tanh(x / scale), fp16 / fp32, in place and out of place, ragged tail."""
    from exllamav3_amd import ext
    rng = np.random.default_rng(5)
    for numel, scale in ((8 * 257 + 3, 30.0), (4096, 50.0), (5, 2.5)):
        x = (rng.standard_normal(numel) * 40).astype(np.float16 if dt == torch.half else np.float32)
        ref = o.softcap(x, scale).astype(np.float32)
        tx = torch.from_numpy(x).to(dev)
        ty = torch.empty_like(tx)
        ext.softcap(tx, ty, scale)
        assert np.allclose(ty.float().cpu().numpy(), ref, rtol=2e-3, atol=2e-3)
        ext.softcap(tx, tx, scale)
        assert torch.equal(tx, ty)


@pytest.mark.parametrize("rows", [1, 7, 160])
def test_linear_wrapper_pad_trim_softcap_post_scale(dev, rows):
    """exllamav3_amd.linear.Linear (modules/linear.py:69-72,561-602): features not multiples of 128 -- input zero-extended to the padded width,
    padded output columns trimmed, softcap and post_scale applied -- against the oracle composition, on the GEMV (rows <= 144) and the
    reconstruct + GEMM (rows > 144) routes."""
    from exllamav3_amd.linear import Linear, LinearEXL3
    k_u, n_u, K, cb = 300, 200, 4, 2                    # padded: 384 x 256
    tr, su, sv = o.synth_linear(384, 256, K, seed=21)
    su[k_u:] = 0                                         # padded input channels carry no weight (modules/linear.py: "the padded weight rows are zeros")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    inner = LinearEXL3(384, 256, T(tr), T(su), T(sv), mcg=False, mul1=True)
    lin = Linear(inner, k_u, n_u, softcap=30.0, post_scale=0.5)
    assert (lin.in_features, lin.out_features) == (384, 256)
    x = np.random.default_rng(rows).standard_normal((rows, k_u)).astype(np.float16)
    y = lin.forward(T(x))
    assert tuple(y.shape) == (rows, n_u) and y.is_contiguous()
    ref = o.linear_wrapper_forward(x, tr, su, sv, K, cb, n_u, softcap_scale=30.0, post_scale=0.5).astype(np.float32)
    err = np.abs(y.float().cpu().numpy() - ref).max() / np.sqrt((ref ** 2).mean())
    assert err < 2e-2, err
    with pytest.raises(RuntimeError):
        Linear(inner, 300, 300)                          # inner op does not have the padded shape


def test_bc_linear_fp16(dev):
    """BC_LinearFP16 (libtorch/linear.h:10-27): y = x @ weight + bias through hgemm."""
    from exllamav3_amd import ext
    g = torch.Generator(device=dev); g.manual_seed(3)
    w = (torch.randn((256, 384), device=dev, generator=g) * 0.1).half()
    b = torch.randn(384, device=dev, generator=g).half()
    for rows in (1, 5):
        x = torch.randn((rows, 256), device=dev, generator=g).half()
        y = torch.empty((rows, 384), dtype=torch.half, device=dev)
        ext.BC_LinearFP16(w, b).run(x, y)
        ref = (x.float() @ w.float() + b.float())
        assert float((y.float() - ref).abs().max()) < 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("gen", [0, 1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("m,k,n", [(256, 128, 256), (300, 512, 512), (1024, 4096, 768), (4096, 1024, 256), (520, 1152, 1024)])
def test_hand_written_nt_mfma_gemm_against_fp32_matmul_and_the_library(dev, m, k, n, gen):
    """Both generations of the hand-written contraction -- exl3_gemm_nt2.hip (gen 2: one wave per SIMD, 128 x 128 wave tiles of v_mfma_f32_32x32x16_f16, the K-loop one
    hand-allocated assembly statement, four LDS buffers filled three K-tiles ahead) and
    exl3_gemm_nt.hip (256 x 256 x 64 tiles, global_load_lds staging with the XOR-permuted source, staggered wave groups): plain store, the fp16
    residual-add epilogue and the fused silu(gate) * up epilogue against an fp32 matmul of the same fp16 operands; ragged row counts (300: the last
    m-tile is partial), one and several K-tiles, strided A (a column range of a wider matrix).  The plain result is also compared with the
    hipBLASLt route (same fp32 accumulation up to summation order)."""
    from exllamav3_amd import ext
    g = torch.Generator(device=dev); g.manual_seed(m + k + n)
    wide = torch.randn((m, k + 64), device=dev, generator=g).half()
    a = wide[:, 32: 32 + k]                                                        # row stride k + 64, 64-byte aligned start
    bt = (torch.randn((n, k), device=dev, generator=g) * 0.05).half()
    ref = a.float() @ bt.float().T
    scale = float(ref.abs().max())
    c = torch.full((m, n), float("nan"), dtype=torch.half, device=dev)
    ext.gemm_nt_mfma(a, bt, c, 0, gen)
    assert bool(torch.isfinite(c).all())
    assert float((c.float() - ref).abs().max()) < 2e-3 * scale
    lib = torch.empty_like(c)
    ext.hgemm_nt(a, bt, lib, route="library")
    assert float((c.float() - lib.float()).abs().max()) < 1e-3 * scale
    # residual add: c = fp16(c + y)
    r0 = torch.randn((m, n), device=dev, generator=g).half()
    r = r0.clone()
    ext.gemm_nt_mfma(a, bt, r, 1, gen)
    assert float((r.float() - (r0.float() + ref)).abs().max()) < 3e-3 * max(scale, 1.0)
    # silu(gate) * up: rows of bt per 256-row tile = 128 gate rows | the 128 up rows of the same outputs
    half_n = n // 2
    gate, up = bt[:half_n], bt[half_n:]
    bt_s = torch.stack([gate.view(half_n // 128, 128, k), up.view(half_n // 128, 128, k)], dim=1).reshape(n, k).contiguous()
    y = torch.full((m, half_n), float("nan"), dtype=torch.half, device=dev)
    ext.gemm_nt_mfma(a, bt_s, y, 2, gen)
    gf = (a.float() @ gate.float().T).half().float(); uf = (a.float() @ up.float().T).half().float()
    yref = gf / (1 + torch.exp(-gf)) * uf
    assert float((y.float() - yref).abs().max()) < 4e-3 * max(float(yref.abs().max()), 1.0)
    with pytest.raises(RuntimeError):
        ext.gemm_nt_mfma(a[:, :96], bt[:, :96].contiguous(), c, 0, gen)            # k % 64 / k % 128


@pytest.mark.parametrize("gen", [0, 4])
@pytest.mark.parametrize("m,k,n", [(300, 192, 384), (256, 64, 128), (1000, 4096, 6144 + 128), (4096, 320, 1280)])
def test_assembly_gemm_narrow_tile_and_odd_shapes(dev, m, k, n, gen):
    """exl3_gemm_nt2.hip outside the 256-multiples: k % 64 (one, three, five K-tiles: the three-stage loop leaves at every tile boundary), n % 128 (the 256 x 128 tile is
    the only form that takes it), ragged m; plain store and the residual add against an fp32 matmul of the same fp16 operands."""
    from exllamav3_amd import ext
    g = torch.Generator(device=dev); g.manual_seed(m * 7 + k + n)
    a = torch.randn((m, k), device=dev, generator=g).half()
    bt = (torch.randn((n, k), device=dev, generator=g) * 0.05).half()
    ref = a.float() @ bt.float().T
    scale = float(ref.abs().max())
    c = torch.full((m, n), float("nan"), dtype=torch.half, device=dev)
    ext.gemm_nt_mfma(a, bt, c, 0, gen)
    assert bool(torch.isfinite(c).all())
    assert float((c.float() - ref).abs().max()) < 2e-3 * scale
    r0 = torch.randn((m, n), device=dev, generator=g).half()
    r = r0.clone()
    ext.gemm_nt_mfma(a, bt, r, 1, gen)
    assert float((r.float() - (r0.float() + ref)).abs().max()) < 3e-3 * max(scale, 1.0)


def test_hgemm_nt_default_route_strided_operands_and_fallbacks(dev):
    """ext.hgemm_nt's default route: W^T and the output as column / row ranges of wider matrices (leading dimensions > k / n) through the own kernel; operands the kernel
    does not take -- a pointer that is not 16-byte aligned, fewer than 256 rows, fp32 output -- silently take the library and give the same values."""
    from exllamav3_amd import ext
    g = torch.Generator(device=dev); g.manual_seed(77)
    m, k, n = 512, 320, 384
    aw = torch.randn((m, k + 64), device=dev, generator=g).half(); a = aw[:, 32: 32 + k]
    bw = (torch.randn((n, k + 128), device=dev, generator=g) * 0.05).half(); bt = bw[:, 64: 64 + k]
    cw = torch.full((m, n + 256), float("nan"), dtype=torch.half, device=dev); c = cw[:, 128: 128 + n]
    ref = a.float() @ bt.float().T
    scale = float(ref.abs().max())
    assert not ext.gemm_nt_own_fills_chip(m, n, a.device) and ext.gemm_nt_own_fills_chip(4096, 4096, a.device) and ext.gemm_nt_own_fills_chip(2048, 4096, a.device) \
        and not ext.gemm_nt_own_fills_chip(1024, 4096, a.device)        # (6 tiles here: the default route hands such a problem to the library; the kernel itself is forced below)
    ext.hgemm_nt(a, bt, c, route="own")
    assert float((c.float() - ref).abs().max()) < 2e-3 * scale
    assert bool(torch.isnan(cw[:, :128]).all()) and bool(torch.isnan(cw[:, 128 + n:]).all())          # nothing written outside the column range
    cd = torch.empty((m, n), dtype=torch.half, device=dev)
    ext.hgemm_nt(a, bt, cd)                                                                              # default route (the library for this size): same values
    assert float((cd.float() - c.float()).abs().max()) < 1e-3 * scale
    lib = torch.empty((m, n), dtype=torch.half, device=dev)
    ext.hgemm_nt(a, bt.contiguous(), lib, route="library")
    assert float((c.float() - lib.float()).abs().max()) < 1e-3 * scale
    # 8-byte aligned start: not a shape of the own kernel (16-byte requests) -> library, same values
    flat = torch.zeros((m * k + 4,), dtype=torch.half, device=dev)
    a8 = flat[4:].view(m, k); a8.copy_(a)
    assert a8.data_ptr() % 16 == 8
    c8 = torch.empty((m, n), dtype=torch.half, device=dev)
    ext.hgemm_nt(a8, bt.contiguous(), c8)
    assert float((c8.float() - ref).abs().max()) < 2e-3 * scale
    # fewer rows than a tile, and fp32 output
    c_small = torch.empty((100, n), dtype=torch.half, device=dev)
    ext.hgemm_nt(a[:100], bt, c_small)
    assert float((c_small.float() - ref[:100]).abs().max()) < 2e-3 * scale
    c32 = torch.empty((m, n), dtype=torch.float, device=dev)
    ext.hgemm_nt(a, bt.contiguous(), c32)
    assert float((c32 - ref).abs().max()) < 2e-3 * scale


@pytest.mark.parametrize("k,n,bounds", [(192, 512, [0, 300, 300, 1000]), (4096, 256, [0, 17, 1041, 2048, 2049]), (64, 768, [0, 256])])
def test_grouped_assembly_gemm_against_fp32_matmul(dev, k, n, bounds):
    """exl3_gemm_nt2_grouped: several problems of one k / n in one launch, row boundaries read from DEVICE memory (an empty problem, partial last tiles, a problem of one
    tile row); the four epilogues -- store, fp16 residual add, silu(gate) * up on interleaved W^T blocks, fp32 output -- against an fp32 matmul per problem; rows past
    rows[count] are left alone."""
    from exllamav3_amd import ext
    g = torch.Generator(device=dev); g.manual_seed(k + n + len(bounds))
    cnt, R = len(bounds) - 1, bounds[-1] + 37                                 # (37 rows of slack behind the last problem: max_rows is an upper bound)
    a = torch.randn((R, k), device=dev, generator=g).half()
    bt = (torch.randn((cnt, n, k), device=dev, generator=g) * 0.05).half()
    rows = torch.tensor(bounds, dtype=torch.int32, device=dev)
    ref = torch.zeros((R, n), dtype=torch.float, device=dev)
    for e in range(cnt):
        ref[bounds[e]: bounds[e + 1]] = a[bounds[e]: bounds[e + 1]].float() @ bt[e].float().T
    scale = float(ref.abs().max())
    used = bounds[-1]
    c = torch.full((R, n), float("nan"), dtype=torch.half, device=dev)
    ext.gemm_nt_grouped(a, bt, c, rows, 0)
    assert bool(torch.isfinite(c[:used]).all()) and bool(torch.isnan(c[used:]).all())
    assert float((c[:used].float() - ref[:used]).abs().max()) < 2e-3 * scale
    c32 = torch.full((R, n), float("nan"), dtype=torch.float, device=dev)
    ext.gemm_nt_grouped(a, bt, c32, rows, 3)
    assert bool(torch.isnan(c32[used:]).all())
    assert float((c32[:used] - ref[:used]).abs().max()) < 1e-3 * scale                       # fp32 out: only the operands are rounded
    assert torch.equal(c32[:used].half(), c[:used])                                           # the fp16 store is the same accumulator, rounded once
    r0 = torch.randn((R, n), device=dev, generator=g).half(); r = r0.clone()
    ext.gemm_nt_grouped(a, bt, r, rows, 1)
    assert torch.equal(r[used:], r0[used:])
    assert float((r[:used].float() - (r0[:used].float() + ref[:used])).abs().max()) < 3e-3 * max(scale, 1.0)
    # silu(gate) * up: per problem, W^T = 128-row blocks of gate and up alternating
    half_n = n // 2
    bt_s = torch.stack([bt[:, :half_n].reshape(cnt, half_n // 128, 128, k), bt[:, half_n:].reshape(cnt, half_n // 128, 128, k)], dim=2).reshape(cnt, n, k).contiguous()
    y = torch.full((R, half_n), float("nan"), dtype=torch.half, device=dev)
    ext.gemm_nt_grouped(a, bt_s, y, rows, 2)
    gf = ref[:, :half_n].half().float(); uf = ref[:, half_n:].half().float()
    yref = gf / (1 + torch.exp(-gf)) * uf
    assert float((y[:used].float() - yref[:used]).abs().max()) < 4e-3 * max(float(yref[:used].abs().max()), 1.0)
    assert bool(torch.isnan(y[used:]).all())


def test_assembly_gemm_random_shapes(dev):
    """exl3_gemm_nt2.hip over 24 seeded random shapes (rows 256 .. 3000, k = 64 * 1 .. 40, n = 128 * 1 .. 24, every loop form incl. forced tiles, plain store / residual add):
    against an fp32 matmul; every exit position of the six- / three- / two-tile loop trips and ragged last row tiles come up."""
    from exllamav3_amd import ext
    rng = np.random.default_rng(2026)
    g = torch.Generator(device=dev); g.manual_seed(11)
    for case in range(24):
        m = int(rng.integers(256, 3001)); k = 64 * int(rng.integers(1, 41)); n = 128 * int(rng.integers(1, 25))
        gen = int(rng.choice([0, 3, 4, 7])) if n % 256 == 0 else int(rng.choice([0, 4]))
        epi = int(rng.integers(0, 2))
        a = torch.randn((m, k), device=dev, generator=g).half()
        bt = (torch.randn((n, k), device=dev, generator=g) / np.sqrt(k)).half()
        ref = a.float() @ bt.float().T
        c0 = torch.randn((m, n), device=dev, generator=g).half() if epi else torch.full((m, n), float("nan"), dtype=torch.half, device=dev)
        c = c0.clone()
        ext.gemm_nt_mfma(a, bt, c, epi, gen)
        want = ref + (c0.float() if epi else 0.0)
        err = float((c.float() - want).abs().max()) / max(float(want.abs().max()), 1.0)
        assert bool(torch.isfinite(c).all()) and err < 3e-3, (case, m, k, n, gen, epi, err)
