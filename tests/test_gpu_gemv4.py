"""Generation-4 GEMV (exl3_gemv4.kspec.hip: 1..4 rows, activation quads in the A-broadcast register layout, no per-wave prologue) against the
oracle (1e-2 * rms, the bar of every GEMV test) and against generation 2 on the same launches (both within the bar of the oracle, and within
2e-3 * rms of each other: same decode, same MFMA order, different grouping of the fp32 partial sums), for every input mode -- raw x, pre-rotated
xh (+ block sums), RMSNorm inside the launch, silu(g) * u from the gate / up slabs -- and output form (final rows, split-k reduce, deferred
slabs finished by a glue kernel)."""
import numpy as np
import pytest
import torch
from oracle import exl3_oracle as o

pytestmark = pytest.mark.gpu
TOL = 1e-2


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _rel(a, ref):
    return float(np.abs(a - ref).max() / np.sqrt((ref ** 2).mean()))


def _gemm(dev, gen4, x, tr, suh, svh, cb, fp32=False, split=0):
    from exllamav3_amd import ext
    ext.set_gemv_gen4(gen4)
    m, n = x.shape[0], svh.shape[0]
    y = torch.full((m, n), float("nan"), dtype=torch.float if fp32 else torch.half, device=dev)
    ext.exl3_gemm(_t(x, dev), _t(tr, dev), y, _t(suh, dev), None, _t(svh, dev), -1, cb == 1, cb == 2, 0, force_split=split)
    ext.set_gemv_gen4(True)
    return y.float().cpu().numpy()


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("cb", [0, 1, 2])
@pytest.mark.parametrize("K", range(1, 9))
def test_gen4_raw_input_all_bitrates_codebooks_rows(dev, K, cb, variant):
    """exl3_gemm, raw x: slices with an odd number of units per wave (3 blocks over 4 waves), a single block, 1..4 rows; final output and split-k."""
    from exllamav3_amd import ext
    ext.set_gemv_variant(variant)
    rng = np.random.default_rng(K * 10 + cb)
    for (k, n, m, split) in [(384, 256, 1, 1), (128, 128, 2, 1), (1024, 128, 3, 2), (896, 256, 4, 1), (2048, 128, 1, 16)]:
        tr, suh, svh = o.synth_linear(k, n, K, seed=K + cb, realistic=True)
        x = rng.standard_normal((m, k)).astype(np.float16)
        ref = o.linear_forward(x, tr, suh, svh, K, cb).astype(np.float32)
        y4 = _gemm(dev, True, x, tr, suh, svh, cb, split=split)
        y2 = _gemm(dev, False, x, tr, suh, svh, cb, split=split)
        assert np.isfinite(y4).all()
        assert _rel(y4, ref) < TOL and _rel(y2, ref) < TOL, (k, n, m, split, _rel(y4, ref), _rel(y2, ref))
        assert _rel(y4, y2) < 3e-3, (k, n, m, split, _rel(y4, y2))


@pytest.mark.parametrize("cb", [0, 2])
@pytest.mark.parametrize("m", [1, 2, 4])
def test_gen4_lm_head_shape_and_long_slices(dev, cb, m):
    """One slice of 32 Hadamard blocks (k = 4096, S = 1: the lm_head launch, 8 waves per workgroup) takes generation 4; 112 blocks in one
    slice (k = 14336, S = 1) is beyond its LDS budget and stays on generation 2: both give the oracle's rows."""
    from exllamav3_amd import ext
    ext.set_gemv_variant(1)
    rng = np.random.default_rng(m)
    for k, n in [(4096, 384), (14336, 128)]:
        tr, suh, svh = o.synth_linear(k, n, 4, seed=5, realistic=True)
        x = rng.standard_normal((m, k)).astype(np.float16)
        ref = o.linear_forward(x, tr, suh, svh, 4, cb, out_fp32=True).astype(np.float32)
        y = _gemm(dev, True, x, tr, suh, svh, cb, fp32=True, split=1)
        assert _rel(y, ref) < TOL


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("cb", [0, 1, 2])
@pytest.mark.parametrize("m", [1, 3, 4])
def test_gen4_rotated_norm_and_act_inputs_through_the_glue_kernels(dev, cb, m, variant):
    """The decode step's launches on a small layer, generation 4 against generation 2 and the oracle:
    q|k|v with the RMSNorm inside the launch (3 matrices, deferred) -> glue_qkv-free check through glue_resid-style reduction (exl3_gemv_ex_norm with
    final outputs), gate|up deferred -> (a) glue_act -> down on the pre-rotated input with block sums, (b) down with silu(g) * u inside the launch."""
    from exllamav3_amd import ext
    ext.set_gemv_variant(variant)
    K, hidden, inter = 4, 512, 896                     # inter / 128 = 7 blocks: odd unit counts per wave in the down launch
    rng = np.random.default_rng(100 * cb + m)
    T = lambda a: _t(a, dev)
    resid = (rng.standard_normal((m, hidden)) * 1.7).astype(np.float16)
    w = (1 + 0.1 * rng.standard_normal(hidden)).astype(np.float16)
    mats = {nm: o.synth_linear(kk, nn, K, seed=sd, realistic=True) for nm, kk, nn, sd in
            [("q", hidden, 256, 1), ("k", hidden, 128, 2), ("v", hidden, 128, 3), ("gate", hidden, inter, 4), ("up", hidden, inter, 5), ("down", inter, hidden, 6)]}
    dm = {nm: tuple(T(a) for a in t) for nm, t in mats.items()}
    xn = o.rms_norm(resid, w, 1e-5)
    ss = torch.empty((m, hidden // 128), dtype=torch.float32, device=dev)
    ext.glue_resid(None, 0, None, None, T(resid), ss, m)                         # per-block sums of squares of the residual (no linear added)
    out = {}
    for gen4 in (True, False):
        ext.set_gemv_gen4(gen4)
        # q|k|v: RMSNorm inside the launch, three matrices, final fp16 outputs
        ys = [torch.full((m, mats[nm][2].shape[0]), float("nan"), dtype=torch.half, device=dev) for nm in ("q", "k", "v")]
        ext.exl3_gemv_ex_norm(T(resid), T(w), ss, 1e-5, [dm[nm][0] for nm in ("q", "k", "v")], ys, [dm[nm][1] for nm in ("q", "k", "v")],
                              [dm[nm][2] for nm in ("q", "k", "v")], m, cb == 1, cb == 2, 0)
        # gate|up: RMSNorm inside the launch, deferred slabs
        gu, S = ext.exl3_gemv_ex_norm(T(resid), T(w), ss, 1e-5, [dm["gate"][0], dm["up"][0]], None, [dm["gate"][1], dm["up"][1]], None, m,
                                      cb == 1, cb == 2, ext.GEMV_OUT_DEFERRED)
        # (a) glue_act -> down on the pre-rotated activation (+ block sums)
        xh = torch.empty((m, inter), dtype=torch.half, device=dev); xs = torch.empty((m, inter // 128), dtype=torch.float32, device=dev)
        a_out = torch.empty((m, inter), dtype=torch.half, device=dev)
        ext.glue_act(gu, S, dm["gate"][2], dm["up"][2], dm["down"][1], xh, xs, m, a_out=a_out)
        d_rot = torch.full((m, hidden), float("nan"), dtype=torch.half, device=dev)
        ext.exl3_gemv_ex(None, [xh], [xs], [dm["down"][0]], [d_rot], None, [dm["down"][2]], m, cb == 1, cb == 2, ext.GEMV_IN_ROTATED)
        # (b) the same from the slabs, silu(g) * u inside the down launch; deferred output finished by glue_resid into a zero residual
        gu2, S2 = ext.exl3_gemv_ex_norm(T(resid), T(w), ss, 1e-5, [dm["gate"][0], dm["up"][0]], None, [dm["gate"][1], dm["up"][1]], None, m,
                                        cb == 1, cb == 2, ext.GEMV_OUT_DEFERRED)
        ds, Sd = ext.exl3_gemv_ex_act(gu2, S2, dm["gate"][2], dm["up"][2], dm["down"][0], None, dm["down"][1], None, m, cb == 1, cb == 2,
                                      ext.GEMV_OUT_DEFERRED)
        r_act = torch.zeros((m, hidden), dtype=torch.half, device=dev); ss2 = torch.empty((m, hidden // 128), dtype=torch.float32, device=dev)
        ext.glue_resid(ds[0], Sd, dm["down"][2], None, r_act, ss2, m)
        torch.cuda.synchronize()
        out[gen4] = [y.float().cpu().numpy() for y in ys] + [a_out.float().cpu().numpy(), d_rot.float().cpu().numpy(), r_act.float().cpu().numpy()]
    ext.set_gemv_gen4(True)
    # oracle
    refs = [o.linear_forward(xn, *mats[nm], K, cb).astype(np.float32) for nm in ("q", "k", "v")]
    g = o.linear_forward(xn, *mats["gate"], K, cb).astype(np.float32); u = o.linear_forward(xn, *mats["up"], K, cb).astype(np.float32)
    a_ref = (g / (1 + np.exp(-g)) * u).astype(np.float16)
    d_ref = o.linear_forward(a_ref, *mats["down"], K, cb).astype(np.float32)
    refs += [a_ref.astype(np.float32), d_ref, d_ref]
    names = ["q", "k", "v", "silu(g)*u", "down (rotated input)", "down (act inside)"]
    for i, nm in enumerate(names):
        tol = 2e-2 if i >= 3 else TOL                    # a product of two linears' outputs / two chained linears
        assert np.isfinite(out[True][i]).all(), nm
        assert _rel(out[True][i], refs[i]) < tol, (nm, _rel(out[True][i], refs[i]))
        assert _rel(out[False][i], refs[i]) < tol, (nm, "generation 2", _rel(out[False][i], refs[i]))
        assert _rel(out[True][i], out[False][i]) < 4e-3, (nm, _rel(out[True][i], out[False][i]))


@pytest.mark.parametrize("m", [1, 4])
def test_gen4_rmsnorm_hidden_8192_and_graph_replay(dev, m):
    """Llama-70B width: 64 per-block sums of squares per row (two 32-lane passes in the task), one Hadamard block per slice; captured into a
    hipGraph and replayed (no host-side state)."""
    from exllamav3_amd import ext
    ext.set_gemv_variant(1)
    k, n, K, cb = 8192, 256, 3, 2
    rng = np.random.default_rng(m)
    tr, suh, svh = o.synth_linear(k, n, K, seed=9, realistic=True)
    w = (1 + 0.1 * rng.standard_normal(k)).astype(np.float16)
    r0 = (rng.standard_normal((m, k)) * 2.5).astype(np.float16)
    T = lambda a: _t(a, dev)
    tr_d, su_d, sv_d, w_d, r_d = T(tr), T(suh), T(svh), T(w), T(r0)
    ss = torch.empty((m, k // 128), dtype=torch.float32, device=dev)
    ext.glue_resid(None, 0, None, None, r_d, ss, m)
    y = torch.full((m, n), float("nan"), dtype=torch.half, device=dev)
    run = lambda: ext.exl3_gemv_ex_norm(r_d, w_d, ss, 1e-5, [tr_d], [y], [su_d], [sv_d], m, False, True, 0)
    run(); torch.cuda.synchronize()
    ref = o.linear_forward(o.rms_norm(r0, w, 1e-5), tr, suh, svh, K, cb).astype(np.float32)
    assert _rel(y.float().cpu().numpy(), ref) < TOL
    y0 = y.clone()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            run()
    y.zero_(); g.replay(); g.replay(); torch.cuda.synchronize()
    assert torch.equal(y, y0)


@pytest.mark.parametrize("cb", [0, 1, 2])
@pytest.mark.parametrize("tokens,m", [(1, 1), (2, 1), (1, 3)])
def test_gen4_table_launches_moe_block_into_fixed_point_accumulator(dev, cb, tokens, m):
    """The MoE block's two indexed launches of the fx pipeline against the oracle: exl3_mgemm_deferred (table mode, raw x, slabs per slot over the
    [gate | up] tables) -> exl3_mgemm_act_fx (silu(g) * u from the slabs of slot j / slot bszm + j, routing weight and svh applied per split-k partial,
    rows ADDED into the int64 accumulator of token j / top_k).  R starts from a known residual; R_after - R_before = sum_k w_k expert_k(x)."""
    from exllamav3_amd import ext
    ext.set_gemv_variant(1)
    K, hidden, inter, E, top = 4, 256, 896, 5, 2
    rng = np.random.default_rng(10 * cb + tokens + m)
    T = lambda a: _t(a, dev)
    gate = [o.synth_linear(hidden, inter, K, seed=10 + e, realistic=True) for e in range(E)]
    up = [o.synth_linear(hidden, inter, K, seed=30 + e, realistic=True) for e in range(E)]
    down = [o.synth_linear(inter, hidden, K, seed=50 + e, realistic=True) for e in range(E)]
    keep = []
    def table(mats, i):
        ts = [T(t[i]) for t in mats]; keep.extend(ts)
        return torch.tensor([t.data_ptr() for t in ts], dtype=torch.long, device=dev)
    gu_B, gu_suh, gu_svh = (table(gate + up, i) for i in range(3))
    d_B, d_suh, d_svh = (table(down, i) for i in range(3))
    sel = np.array([[3, 0], [1, 3]][:tokens], dtype=np.int64)                 # (tokens, top)
    w = np.array([[0.7, 0.3], [0.45, 0.55]][:tokens], dtype=np.float16)
    x = rng.standard_normal((tokens, m, hidden)).astype(np.float16)
    sel2 = np.concatenate([sel.reshape(-1), sel.reshape(-1) + E])            # [gate slots | up slots]
    A = x if tokens == 1 else np.concatenate([np.repeat(x, top, axis=0)] * 2) # one shared row set, or one per slot
    r0 = (rng.standard_normal((tokens * m, hidden)) * 3).astype(np.float32)
    R = torch.from_numpy(np.rint(r0.astype(np.float64) * 2.0 ** 32).astype(np.int64)).to(dev)
    R0 = R.clone()
    slab, S = ext.exl3_mgemm_deferred(T(A), gu_B, gu_suh, T(sel2), K, cb == 1, cb == 2, inter)
    ext.exl3_mgemm_act_fx(slab, S, gu_svh, E, d_B, d_suh, d_svh, T(sel.reshape(-1)), T(w.reshape(-1)), R, K, cb == 1, cb == 2, inter, m=m, num_tokens=tokens)
    torch.cuda.synchronize()
    got = ((R - R0).double() / 2.0 ** 32).cpu().numpy().reshape(tokens, m, hidden)
    lin = lambda xx, t, fp32=False: o.linear_forward(xx, t[0], t[1], t[2], K, cb, out_fp32=fp32).astype(np.float32)
    for t in range(tokens):
        ref = np.zeros((m, hidden), dtype=np.float32)
        for j in range(top):
            e = int(sel[t, j])
            g, u = lin(x[t], gate[e]), lin(x[t], up[e])
            a = (g / (1 + np.exp(-g)) * u).astype(np.float16)
            ref += float(w[t, j]) * lin(a, down[e], fp32=True)
        assert _rel(got[t], ref) < 2e-2, (t, _rel(got[t], ref))
    # order independence: a second run from the same start gives the same integers
    R2 = R0.clone()
    slab, S = ext.exl3_mgemm_deferred(T(A), gu_B, gu_suh, T(sel2), K, cb == 1, cb == 2, inter)
    ext.exl3_mgemm_act_fx(slab, S, gu_svh, E, d_B, d_suh, d_svh, T(sel.reshape(-1)), T(w.reshape(-1)), R2, K, cb == 1, cb == 2, inter, m=m, num_tokens=tokens)
    assert torch.equal(R, R2)


def test_routing_on_the_fixed_point_residual_matches_routing_std_norm(dev):
    """routing_std_fx (router input = RMSNorm of the int64 fixed-point residual, exact sums of squares taken inside the launch) against
    routing_std_norm on the fp16 copy of the same residual with glue_resid's block sums: same experts, weights, normalised row and block sums."""
    from exllamav3_amd import ext
    E, top = 8, 2
    rng = np.random.default_rng(3)
    T = lambda a: _t(a, dev)
    for bsz, hidden in ((1, 1024), (3, 1024), (2, 5120)):                    # hidden > 4096: the row does not stay in registers
        xh = (rng.standard_normal((bsz, hidden)) * 2).astype(np.float16)
        R = torch.from_numpy(np.rint(xh.astype(np.float64) * 2.0 ** 32).astype(np.int64)).to(dev)
        w = T((1 + 0.1 * rng.standard_normal(hidden)).astype(np.float16)); gate = T((rng.standard_normal((hidden, E)) * 0.05).astype(np.float16))
        mk = lambda: (torch.empty((bsz, E), dtype=torch.half, device=dev), torch.empty((bsz, top), dtype=torch.long, device=dev),
                      torch.empty((bsz, top), dtype=torch.half, device=dev), torch.empty((2, bsz * top), dtype=torch.long, device=dev),
                      torch.empty((bsz, hidden), dtype=torch.half, device=dev))
        sc0, sel0, w0, gu0, xn0 = mk(); sc1, sel1, w1, gu1, xn1 = mk()
        ss = torch.empty((bsz, hidden // 128), dtype=torch.float32, device=dev); ss1 = torch.full_like(ss, float("nan"))
        x_d = T(xh)
        ext.glue_resid(None, 0, None, None, x_d, ss, bsz)
        ext.routing_std_norm(x_d, w, ss, 1e-5, xn0, gate, sc0, sel0, w0, gu_slots=gu0)
        ext.routing_std_fx(R, w, ss1, 1e-5, xn1, gate, sc1, sel1, w1, gu_slots=gu1)
        torch.cuda.synchronize()
        assert torch.equal(sel0, sel1) and torch.equal(gu0, gu1) and torch.equal(xn0, xn1) and torch.equal(w0, w1) and torch.equal(sc0, sc1)
        assert np.allclose(ss1.cpu().numpy(), ss.cpu().numpy(), rtol=1e-6)


@pytest.mark.parametrize("m,hidden", [(1, 4096), (3, 1024), (2, 8192)])
def test_merged_step_boundaries_equal_the_two_launch_forms(dev, m, hidden):
    """fx_init_prep == fx_init + qkv_prep and fx_finish_rotate == fx_finish + glue_rotate, bit for bit (one kernel boundary less each per decode step;
    hidden 8192: a half-wave walks two Hadamard blocks of its row)."""
    from exllamav3_amd import ext
    rng = np.random.default_rng(m + hidden)
    T = lambda a: _t(a, dev)
    x = T((rng.standard_normal((m, hidden)) * 2).astype(np.float16))
    hd, page, pps = 128, 256, 4
    inv_freq = T((1.0 / (500000.0 ** (np.arange(0, hd, 2, dtype=np.float32) / hd))).astype(np.float32))
    pos = T(np.array([5, 300, 777][:m], dtype=np.int32)); bt = torch.arange(m * pps, dtype=torch.int32, device=dev).view(m, pps)
    mk = lambda: (torch.zeros((m, hidden), dtype=torch.long, device=dev), torch.zeros((m, hidden // 128), dtype=torch.float32, device=dev),
                  torch.zeros((m, 64), dtype=torch.float32, device=dev), torch.zeros((m, 64), dtype=torch.float32, device=dev),
                  torch.zeros((m,), dtype=torch.long, device=dev))
    R0, ss0, sn0, cs0, sl0 = mk(); R1, ss1, sn1, cs1, sl1 = mk()
    ext.fx_init(x, R0, ss0, m); ext.qkv_prep(inv_freq, pos, hd, bt, page, sn0, cs0, sl0)
    ext.fx_init_prep(x, R1, ss1, m, inv_freq, pos, hd, bt, page, sn1, cs1, sl1)
    assert torch.equal(R0, R1) and torch.equal(ss0, ss1) and torch.equal(sn0, sn1) and torch.equal(cs0, cs1) and torch.equal(sl0, sl1)
    # a residual that is not exactly representable in fp16
    R0 += torch.randint(-2 ** 30, 2 ** 30, R0.shape, device=dev)
    w = T((1 + 0.1 * rng.standard_normal(hidden)).astype(np.float16)); suh = T(np.where(rng.random(hidden) < 0.5, -1.0, 1.0).astype(np.float16))
    xa = torch.empty((m, hidden), dtype=torch.half, device=dev); ssa = torch.empty((m, hidden // 128), dtype=torch.float32, device=dev)
    xha = torch.empty_like(xa); xsa = torch.empty_like(ssa)
    ext.fx_finish(R0, xa, ssa, m); ext.glue_rotate(xa, ssa, w, 1e-5, [suh], [xha], m, xsums=[xsa])
    xb = torch.empty_like(xa); ssb = torch.empty_like(ssa); xhb = torch.full_like(xa, float("nan")); xsb = torch.empty_like(ssa)
    ext.fx_finish_rotate(R0, xb, ssb, w, 1e-5, suh, xhb, xsb, m)
    assert torch.equal(xa, xb) and torch.equal(ssa, ssb) and torch.equal(xha, xhb) and torch.equal(xsa, xsb)
