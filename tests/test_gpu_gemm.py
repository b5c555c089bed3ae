"""GPU parity of the fused EXL3 GEMV / small-m GEMM (decode path) against the oracle, through the C-ABI.

Tolerance: max |y - y_ref| <= 1e-2 * rms(y_ref)   (rtol ~ 1e-2 on outputs of RMS ~ 1; the reference's
own kernel-vs-reconstruct test uses rtol = atol = 0.05, tests/test_qgemm.py:31-58)."""
import numpy as np
import pytest
import torch
from oracle import exl3_oracle as o

pytestmark = pytest.mark.gpu
TOL = 1e-2


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _run(dev, k, n, K, cb, m, variant, out_fp32=False, force_split=0, bias=False, realistic=False, seed=None):
    from exllamav3_amd import ext
    ext.set_gemv_variant(variant)
    tr, suh, svh = o.synth_linear(k, n, K, seed=seed, realistic=realistic)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((m, k)).astype(np.float16)
    b = (rng.standard_normal(n) * 0.5).astype(np.float16) if bias else None
    ref = o.linear_forward(x, tr, suh, svh, K, cb, bias=b, out_fp32=out_fp32).astype(np.float32)
    y = torch.full((m, n), float("nan"), dtype=torch.float if out_fp32 else torch.half, device=dev)
    rc = ext.exl3_gemm(_t(x, dev), _t(tr, dev), y, _t(suh, dev), None, _t(svh, dev), -1, cb == 1, cb == 2, 0,
                       bias=None if b is None else _t(b, dev), force_split=force_split)
    assert rc >= 1
    got = y.float().cpu().numpy()
    assert np.isfinite(got).all()
    err = np.abs(got - ref).max() / np.sqrt((ref ** 2).mean())
    return err


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("cb", [0, 1, 2])
@pytest.mark.parametrize("K", range(1, 9))
def test_gemv_all_bitrates(dev, K, cb, variant):
    assert _run(dev, 512, 256, K, cb, 1, variant) < TOL
    assert _run(dev, 1024, 128, K, cb, 3, variant) < TOL


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 8, 9, 15, 16, 17, 31, 33])
def test_gemm_batch_sizes(dev, m, variant):
    # tests/test_qgemm.py bs list (those <= 33); m > 16 runs 16-row passes
    for cb in (0, 2):
        assert _run(dev, 1024, 384, 4, cb, m, variant) < TOL


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("split", [1, 2, 3, 4, 8, 16])
def test_split_k(dev, split, variant):
    # includes uneven slices and slices shorter than 4 Hadamard blocks (some waves of a workgroup get no work)
    for cb in (0, 1, 2):
        assert _run(dev, 2048, 256, 4, cb, 2, variant, force_split=split) < TOL
        assert _run(dev, 2048, 256, 3, cb, 5, variant, out_fp32=True, force_split=split) < TOL
        assert _run(dev, 1664, 128, 4, cb, 1, variant, force_split=split) < TOL       # 13 blocks


@pytest.mark.parametrize("variant", [0, 1])
def test_fp32_out_bias_realistic(dev, variant):
    for cb in (0, 1, 2):
        assert _run(dev, 512, 512, 4, cb, 4, variant, out_fp32=True, bias=True) < TOL
        assert _run(dev, 512, 512, 5, cb, 1, variant, out_fp32=False, bias=True) < TOL
        assert _run(dev, 4096, 1024, 4, cb, 1, variant, realistic=True) < TOL


@pytest.mark.parametrize("variant", [0, 1])
def test_llama_shapes(dev, variant):
    # science/qgemm_benchmark.py shapes at 4 bpw (sizes the oracle finishes in seconds)
    for (k, n) in [(4096, 4096), (4096, 1024), (14336, 4096)]:
        assert _run(dev, k, n, 4, 2, 1, variant) < TOL
    assert _run(dev, 4096, 14336, 4, 0, 16, variant) < TOL


def test_exact_variant_matches_reconstruct_matmul_tightly(dev):
    """EXACT variant feeds the reference's fp16 weights bit-for-bit to the MFMA: agreement with an fp64 matmul over the
    reconstructed weights is limited only by the fp16 rounding of the output."""
    assert _run(dev, 1024, 256, 4, 0, 4, 0, out_fp32=True) < 2e-4
    assert _run(dev, 1024, 256, 4, 2, 4, 0, out_fp32=True) < 2e-4


def test_mgemm_broadcast(dev):
    from exllamav3_amd import ext
    ext.set_gemv_variant(1)
    k, K, cb, m = 1024, 4, 2, 2
    ns = [512, 128, 128]
    rng = np.random.default_rng(0)
    x = rng.standard_normal((m, k)).astype(np.float16)
    mats = [o.synth_linear(k, n, K, seed=50 + i) for i, n in enumerate(ns)]
    Bs = [_t(t[0], dev) for t in mats]; su = [_t(t[1], dev) for t in mats]; sv = [_t(t[2], dev) for t in mats]
    Cs = [torch.full((m, n), float("nan"), dtype=torch.half, device=dev) for n in ns]
    for split in (0, 1, 4):
        ext.exl3_mgemm_bcast(_t(x, dev), Bs, Cs, su, sv, False, True, force_split=split)
        for (tr, suh, svh), C in zip(mats, Cs):
            ref = o.linear_forward(x, tr, suh, svh, K, cb).astype(np.float32)
            got = C.float().cpu().numpy()
            assert np.abs(got - ref).max() / np.sqrt((ref ** 2).mean()) < TOL


def test_bc_linear_exl3(dev):
    from exllamav3_amd import ext
    k, n, K = 512, 256, 4
    tr, suh, svh = o.synth_linear(k, n, K)
    x = np.random.default_rng(0).standard_normal((1, 3, k)).astype(np.float16)
    bc = ext.BC_LinearEXL3(_t(tr, dev), _t(suh, dev), _t(svh, dev), K, None, False, False, None)
    y = bc.run_alloc(_t(x, dev), n, True)
    assert y.shape == (1, 3, n) and y.dtype == torch.float
    ref = o.linear_forward(x.reshape(3, k), tr, suh, svh, K, 0, out_fp32=True)
    assert np.abs(y.cpu().numpy().reshape(3, n) - ref).max() / np.sqrt((ref ** 2).mean()) < TOL


def test_graph_capture_replay(dev):
    """Decode-step launches must be capturable into a hipGraph (reference: graph.cuh:100-137)."""
    from exllamav3_amd import ext
    k, n, K = 1024, 512, 4
    tr, suh, svh = o.synth_linear(k, n, K)
    x = _t(np.random.default_rng(0).standard_normal((1, k)).astype(np.float16), dev)
    B, su, sv = _t(tr, dev), _t(suh, dev), _t(svh, dev)
    y = torch.zeros((1, n), dtype=torch.half, device=dev)
    ext.exl3_gemm(x, B, y, su, None, sv, -1, False, False, 0)
    torch.cuda.synchronize()
    expect = y.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        y.zero_()
        with torch.cuda.graph(g, stream=s):
            ext.exl3_gemm(x, B, y, su, None, sv, -1, False, False, 0)
    y.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, expect)


def test_two_streams_share_the_device_workspace_safely(dev):
    """The split-k slabs live in one per-device workspace: when the issuing stream changes the library orders the new stream behind the old one, so
    GEMVs issued alternately on two streams (no host synchronisation in between) still give the oracle's rows; switching streams in the middle of a
    graph capture is refused."""
    from exllamav3_amd import ext
    k, n, K, cb = 4096, 512, 4, 2
    rng = np.random.default_rng(5)
    tr, suh, svh = o.synth_linear(k, n, K, seed=3, realistic=True)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tr_d, su_d, sv_d = T(tr), T(suh), T(svh)
    xs = [rng.standard_normal((1, k)).astype(np.float16) for _ in range(8)]
    xd = [T(x) for x in xs]
    ys = [torch.full((1, n), float("nan"), dtype=torch.half, device=dev) for _ in range(8)]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for i in range(8):
        with torch.cuda.stream(s1 if i % 2 == 0 else s2):
            ext.exl3_gemm(xd[i], tr_d, ys[i], su_d, None, sv_d, -1, False, True, 0, force_split=8)
    torch.cuda.synchronize()
    for i in range(8):
        ref = o.linear_forward(xs[i], tr, suh, svh, K, cb).astype(np.float32)
        assert np.abs(ys[i].float().cpu().numpy() - ref).max() < 1e-2 * np.sqrt((ref ** 2).mean()), i
    # a stream change while the previous stream is capturing cannot be ordered: refused
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            ext.exl3_gemm(xd[0], tr_d, ys[0], su_d, None, sv_d, -1, False, True, 0, force_split=8)
            with torch.cuda.stream(s2):
                with pytest.raises(RuntimeError):
                    ext.exl3_gemm(xd[1], tr_d, ys[1], su_d, None, sv_d, -1, False, True, 0, force_split=8)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    ref = o.linear_forward(xs[0], tr, suh, svh, K, cb).astype(np.float32)
    assert np.abs(ys[0].float().cpu().numpy() - ref).max() < 1e-2 * np.sqrt((ref ** 2).mean())


@pytest.mark.parametrize("max_waves", [1, 3, 4, 7, 16])
def test_gen2_waves_per_workgroup(dev, max_waves):
    """gen 2 lets up to 16 waves split a workgroup's k-slice (uneven block counts, waves without work)."""
    from exllamav3_amd import ext
    ext.set_gemv_max_waves(max_waves)
    try:
        for (k, n, m, split) in [(4096, 256, 1, 1), (1664, 128, 2, 1), (14336, 128, 1, 0), (2048, 384, 16, 2), (1024, 128, 9, 1)]:
            for cb in (0, 2):
                assert _run(dev, k, n, 4, cb, m, 1, force_split=split) < TOL
        assert _run(dev, 4096, 256, 3, 1, 4, 0, force_split=1) < TOL
    finally:
        ext.set_gemv_max_waves(0)


@pytest.mark.parametrize("cb", [0, 2])
@pytest.mark.parametrize("k,n,K", [(512, 256, 4), (1024, 384, 3)])
def test_mgemm_indexed_moe_forms(dev, cb, k, n, K):
    """exl3_mgemm with pointer tables (quant/exl3_gemm.cuh:58-78): (1) shared input, indices -> per-slot outputs (gate/up of selected
    experts); (2) per-slot inputs + routing weights -> weighted sum into C[0] (down); (3) expert-range filter (compaction, re-basing,
    untouched slots); (4) num_tokens > 1 grouped reduction.  Against the oracle linear per slot."""
    from exllamav3_amd import ext
    E, top, m = 6, 3, 2
    rng = np.random.default_rng(k + n + K + cb)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    mats = [o.synth_linear(k, n, K, seed=100 + e, realistic=True) for e in range(E)]
    tB = [T(t[0]) for t in mats]; tsu = [T(t[1]) for t in mats]; tsv = [T(t[2]) for t in mats]
    ptr = lambda ts: torch.tensor([t.data_ptr() for t in ts], dtype=torch.long, device=dev)
    pB, psu, psv = ptr(tB), ptr(tsu), ptr(tsv)
    lin = lambda x, e, fp32=False: o.linear_forward(x, mats[e][0], mats[e][1], mats[e][2], K, cb, out_fp32=fp32).astype(np.float32)
    tol = lambda ref: 1e-2 * np.sqrt((ref ** 2).mean()) + 1e-3
    # (1) shared input
    x = rng.standard_normal((1, m, k)).astype(np.float16)
    sel = np.array([4, 0, 3], dtype=np.int64)
    C = torch.full((top, m, n), float("nan"), dtype=torch.half, device=dev)
    ext.exl3_mgemm(T(x), pB, C, psu, None, psv, T(sel), None, K, -1, cb == 1, cb == 2, -1, -1, 0)
    for j, e in enumerate(sel):
        ref = lin(x[0], int(e))
        assert np.abs(C[j].float().cpu().numpy() - ref).max() < tol(ref)
    # (2) per-slot inputs + weights, fp32 and fp16 outputs
    xs = rng.standard_normal((top, m, k)).astype(np.float16)
    w = np.array([0.5, 0.3, 0.2], dtype=np.float16)
    for dt in (torch.float, torch.half):
        C = torch.zeros((top, m, n), dtype=dt, device=dev)
        ext.exl3_mgemm(T(xs), pB, C, psu, None, psv, T(sel), T(w), K, -1, cb == 1, cb == 2, -1, -1, 0)
        ref = sum(float(w[j]) * lin(xs[j], int(e), fp32=True) for j, e in enumerate(sel))
        assert np.abs(C[0].float().cpu().numpy() - ref).max() < 2 * tol(ref)
    # (3) expert range [3, 6): slots with experts 4 and 3 run (compacted to slots 0, 1 and re-based to a local table), slot 2 untouched
    lo, hi = 3, 6
    pBl, psul, psvl = ptr(tB[lo:hi]), ptr(tsu[lo:hi]), ptr(tsv[lo:hi])
    C = torch.full((top, m, n), 7.0, dtype=torch.half, device=dev)
    ext.exl3_mgemm(T(x), pBl, C, psul, None, psvl, T(sel), None, K, -1, cb == 1, cb == 2, lo, hi, 0)
    kept = [int(e) for e in sel if lo <= e < hi]
    for j, e in enumerate(kept):
        ref = lin(x[0], e)
        assert np.abs(C[j].float().cpu().numpy() - ref).max() < tol(ref)
    assert bool((C[len(kept):] == 7.0).all())
    # (4) two tokens x two experts each, grouped reduction into rows 0 and 1
    sel2 = np.array([1, 5, 2, 0], dtype=np.int64); w2 = np.array([0.6, 0.4, 0.7, 0.3], dtype=np.float16)
    xs2 = rng.standard_normal((4, 1, k)).astype(np.float16)
    C = torch.zeros((4, 1, n), dtype=torch.float, device=dev)
    ext.exl3_mgemm(T(xs2), pB, C, psu, None, psv, T(sel2), T(w2), K, -1, cb == 1, cb == 2, -1, -1, 0, num_tokens=2)
    for t in range(2):
        ref = sum(float(w2[2 * t + j]) * lin(xs2[2 * t + j], int(sel2[2 * t + j]), fp32=True) for j in range(2))
        assert np.abs(C[t].float().cpu().numpy() - ref).max() < 2 * tol(ref)
    with pytest.raises(RuntimeError):
        ext.exl3_mgemm(T(xs2), pB, C, psu, None, psv, T(sel2), T(w2), K, -1, cb == 1, cb == 2, 0, 4, 0, num_tokens=2)
    # (5) weights AND an expert range (the down launch of an expert-parallel rank): the in-range slots are compacted to the front together with their
    # inputs' slots and weights, and only THEY are summed into C[0] (exl3_gemm_kernel.cuh:101-127 continues with bszm = the in-range count) -- a C full
    # of NaN needs no zero fill; a token none of whose experts live on this rank gives zeros
    for dt in (torch.float, torch.half):
        C = torch.full((top, m, n), float("nan"), dtype=dt, device=dev)
        ext.exl3_mgemm(T(xs), pBl, C, psul, None, psvl, T(sel), T(w), K, -1, cb == 1, cb == 2, lo, hi, 0)
        kept_j = [j for j, e in enumerate(sel) if lo <= e < hi]
        # the kernel compacts indices and weights; slot j' of the compacted list reads input slot j' (per-slot inputs are the caller's compacted
        # activations: the gate / up launches of the same rank wrote them in the same compacted order)
        ref = sum(float(w[j]) * lin(xs[jc], int(sel[j]), fp32=True) for jc, j in enumerate(kept_j))
        got = C[0].float().cpu().numpy()
        assert np.isfinite(got).all() and np.abs(got - ref).max() < 2 * tol(ref)
        C = torch.full((top, m, n), float("nan"), dtype=dt, device=dev)
        ext.exl3_mgemm(T(xs), pBl, C, psul, None, psvl, T(np.array([0, 1, 2], np.int64)), T(w), K, -1, cb == 1, cb == 2, lo, hi, 0)
        assert bool((C[0] == 0).all())


@pytest.mark.parametrize("cb,K", [(0, 4), (2, 3)])
@pytest.mark.parametrize("m", [1, 3, 7, 21])
def test_mgemm_per_matrix_widths(dev, cb, K, m):
    """exl3_mgemm with size_n_list / c_ptrs (quant/exl3_gemm.cu:433-447, the x-side fan of libtorch/dsv4_attn.cpp:88-98): matrices of different widths in
    one launch, every output at its own address with its own row stride; without indices (slot j = matrix j) and with them (a selection, in another
    order); fp16 and fp32.  Against the oracle linear per matrix; nothing is written past a matrix's width or into C."""
    from exllamav3_amd import ext
    k = 1024
    widths = [384, 128, 640, 256]
    rng = np.random.default_rng(7 * m + K + cb)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    mats = [o.synth_linear(k, n, K, seed=300 + i, realistic=True) for i, n in enumerate(widths)]
    tB = [T(t[0]) for t in mats]; tsu = [T(t[1]) for t in mats]; tsv = [T(t[2]) for t in mats]
    ptr = lambda ts: torch.tensor([t.data_ptr() for t in ts], dtype=torch.long, device=dev)
    pB, psu, psv = ptr(tB), ptr(tsu), ptr(tsv)
    nl = torch.tensor(widths, dtype=torch.int32, device=dev)
    x = rng.standard_normal((1, m, k)).astype(np.float16)
    tol = lambda ref: 1e-2 * np.sqrt((ref ** 2).mean()) + 1e-3
    for dt in (torch.half, torch.float):
        for sel in (None, [2, 0, 3]):
            outs = [torch.full((m, n + 128), 7.0, dtype=dt, device=dev) for n in widths]        # 128 guard columns behind every output
            flat = [t.view(-1) for t in outs]
            Cd = torch.full((1, m, max(widths)), 9.0, dtype=dt, device=dev)                     # dtype + max-width carrier: never written
            idx = None if sel is None else T(np.array([sel], dtype=np.int64))
            ext.exl3_mgemm(T(x), pB, Cd, psu, None, psv, idx, None, K, -1, cb == 1, cb == 2, -1, -1, 0, 1, nl, ptr(flat))
            ran = range(len(widths)) if sel is None else sel
            for i, n in enumerate(widths):
                got = flat[i][: m * n].view(m, n).float().cpu().numpy()
                if i in ran:
                    ref = o.linear_forward(x[0], mats[i][0], mats[i][1], mats[i][2], K, cb, out_fp32=dt == torch.float).astype(np.float32)
                    assert np.abs(got - ref).max() < tol(ref), (dt, sel, i)
                else:
                    assert bool((flat[i] == 7.0).all())
                assert bool((flat[i][m * n:] == 7.0).all())
            assert bool((Cd == 9.0).all())
    with pytest.raises(RuntimeError):
        ext.exl3_mgemm(T(x), pB, Cd, psu, None, psv, None, None, K, -1, cb == 1, cb == 2, -1, -1, 0, 1, nl, None)


# ---- generation 3 (exl3_gemm3.kspec.hip): 9..64 rows per pass through the LDS transpose into 16x16x32 MFMAs ----------------------------

@pytest.mark.parametrize("cb", [0, 1, 2])
@pytest.mark.parametrize("K", range(1, 9))
def test_gemm3_all_bitrates(dev, K, cb):
    from exllamav3_amd import ext
    ext.set_gemm3_min_rows(5)
    assert _run(dev, 512, 256, K, cb, 12, 1) < TOL
    assert _run(dev, 1024, 128, K, cb, 32, 1) < TOL
    assert _run(dev, 640, 128, K, cb, 50, 1) < TOL            # 5 Hadamard blocks (odd: ring slot swap), 64-row pass


@pytest.mark.parametrize("m", [9, 16, 17, 24, 32, 33, 47, 64, 65, 144])
@pytest.mark.parametrize("split", [0, 1, 3, 8])
def test_gemm3_rows_and_splits(dev, m, split):
    # uneven slices, several activation chunks per slice (k = 2816 at split 1: 22 Hadamard blocks against a 2..4 block LDS budget),
    # tails of 1 row (33, 65) that fall back to generation 2 inside the same call
    from exllamav3_amd import ext
    ext.set_gemm3_min_rows(5)
    for cb in (0, 2):
        assert _run(dev, 2816, 256, 4, cb, m, 1, force_split=split) < TOL
        assert _run(dev, 1024, 384, 3, cb, m, 1, out_fp32=True, bias=True, force_split=split, realistic=True, seed=5) < TOL


def test_gemm3_agrees_with_generation_2(dev):
    # same products, fp32 accumulation in a different order: the two kernels agree far inside the oracle tolerance
    from exllamav3_amd import ext
    tr, suh, svh = o.synth_linear(4096, 512, 4, seed=11, realistic=True)
    x = torch.randn((16, 4096), device=dev).half()
    ys = []
    for mr in (0, 5):
        ext.set_gemm3_min_rows(mr)
        y = torch.empty((16, 512), dtype=torch.float, device=dev)
        ext.exl3_gemm(x, _t(tr, dev), y, _t(suh, dev), None, _t(svh, dev), -1, False, True, 0)
        ys.append(y)
    ext.set_gemm3_min_rows(5)
    err = (ys[0] - ys[1]).abs().max() / ys[0].pow(2).mean().sqrt()
    assert err < 2e-3, err


@pytest.mark.parametrize("m", [5, 8, 9, 16])
@pytest.mark.parametrize("split", [1, 2, 5])
def test_gemm3_rotated_input_long_slices(dev, m, split):
    """Pre-rotated activations (the fused pipeline's route) over slices much longer than one LDS chunk: 32 Hadamard blocks at split 1
    against a chunk of <= 8, with the next chunk's rows prefetched underneath the streaming loop; against the raw-input route."""
    from exllamav3_amd import ext
    k, n, K = 4096, 256, 4
    tr, suh, svh = o.synth_linear(k, n, K, seed=21, realistic=True)
    ttr, tsu, tsv = _t(tr, dev), _t(suh, dev), _t(svh, dev)
    x = torch.randn((m, k), device=dev, generator=torch.Generator(device=dev).manual_seed(m)).half()
    xh = torch.empty_like(x)
    ext.had_r_128(x, xh, tsu, None, 1.0)
    y_rot = torch.full((m, n), float("nan"), dtype=torch.half, device=dev)
    ext.exl3_gemv_ex(None, [xh], None, [ttr], [y_rot], None, [tsv], m, False, True, ext.GEMV_IN_ROTATED, force_split=split)
    ref = o.linear_forward(x.cpu().numpy(), tr, suh, svh, K, 2).astype(np.float32)
    got = y_rot.float().cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() / np.sqrt((ref ** 2).mean()) < TOL


def test_gemm3_lm_head_width_at_64_rows_and_oversized_forced_split(dev):
    """at 33..64 rows an lm_head-wide n (1002 column blocks) made the generation-3 split search reject every candidate -- the slab
    workspace test was applied to the one-slice launch that writes no slabs -- and the slice length was then divided by a zero split (SIGFPE on the host);
    a forced split above the search's 64-slice cap ended the same way.  Both now run and match the oracle (sampled columns for the wide one)."""
    from exllamav3_amd import ext
    from test_gpu_fullsize import oracle_linear
    ext.set_gemv_variant(1)
    k, n, K, cb, m = 256, 128256, 4, 2, 64
    tr, suh, svh = o.synth_linear(k, n, K, seed=5)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((m, k)).astype(np.float16)
    y = torch.full((m, n), float("nan"), dtype=torch.half, device=dev)
    rc = ext.exl3_gemm(_t(x, dev), _t(tr, dev), y, _t(suh, dev), None, _t(svh, dev), -1, False, True, 0)
    assert rc >= 1
    got = y.float().cpu().numpy()
    assert np.isfinite(got).all()
    cols = [(0, 256), (64000, 64256), (n - 128, n)]
    ref = oracle_linear(x, tr, suh, svh, K, cb, cols=cols).astype(np.float32)
    g = np.concatenate([got[:, a:b] for a, b in cols], axis=-1)
    assert np.abs(g - ref).max() / np.sqrt((ref ** 2).mean()) < TOL
    # forced split beyond 64 slices (96 Hadamard blocks): normalised, not a crash
    assert _run(dev, 96 * 128, 256, 4, 2, 8, 1, force_split=80) < TOL
