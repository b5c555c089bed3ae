"""Host logic of the persistent decode step's planner (exl3_pstep.hip: plan_op / fill_tiles through the C ABI, no GPU): for every op of the BASELINE
shapes and a few odd ones the rectangles of a chip PARTITION the op -- every (column block, 128-row Hadamard block) of every matrix lies in exactly one
rectangle -- and obey what the kernel assumes: at most 12 column blocks per rectangle (a streaming wave's run crosses one column-block boundary at most),
at most 8 blocks per slice where the input is finished from slab lines (one task per service half-wave), at most 4 owned blocks where the op adds into
the residual row, one slice for the lm_head, slices numbered 0..S-1 per column group, K / V side tasks on distinct workgroups."""
import ctypes
import numpy as np
import pytest

SHAPES = {"llama-3.1-8b": (4096, 14336, 32, 8, 128, 128256), "llama-3.2-1b": (2048, 8192, 32, 8, 64, 128256),
          "tiny": (256, 512, 4, 2, 128, 384), "odd": (1024, 2816, 8, 8, 128, 3072), "hd64": (512, 1536, 8, 2, 64, 1024),
          # other public Llama-shaped checkpoints (slice counts 6 .. 25 on 256 CUs; the GPU side: test_gpu_pstep.MID_SHAPES)
          "llama-3.2-3b": (3072, 8192, 24, 8, 128, 128256), "qwen2.5-7b": (3584, 18944, 28, 4, 128, 152064), "qwen2.5-1.5b": (1536, 8960, 12, 2, 128, 151936),
          "llama-2-7b": (4096, 11008, 32, 32, 128, 32000), "tinyllama": (2048, 5632, 32, 4, 64, 32000), "hidden-2560": (2560, 6912, 20, 4, 128, 50304)}


@pytest.mark.parametrize("ncu", [256, 304, 64])
@pytest.mark.parametrize("name", list(SHAPES))
def test_rectangles_partition_every_op(name, ncu):
    from exllamav3_amd import _lib
    l = _lib.lib()
    hidden, inter, hq, hkv, hd, vocab = SHAPES[name]
    qdim, kvdim = hq * hd, hkv * hd
    ops = {0: ([qdim // 128, kvdim // 128, kvdim // 128], hidden // 128), 1: ([hidden // 128], qdim // 128), 2: ([inter // 128, inter // 128], hidden // 128),
           3: ([hidden // 128], inter // 128), 4: ([vocab // 128], hidden // 128)}
    for kind, (ncbs, nblk) in ops.items():
        tiles = np.zeros((ncu, 12), dtype=np.int32); S = ctypes.c_int(0)
        rc = l.exl3_pstep_plan_tiles(hidden, inter, hq, hkv, hd, vocab, ncu, kind, tiles.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.byref(S))
        if rc != 0:
            # legitimate refusals, small chips only: an lm_head wider than 12 column blocks per CU (a streaming wave's run would cross two boundaries), or an
            # op that adds into the residual row whose <= 8-block slices leave column groups wider than the 4 blocks an owner gathers
            assert ncu < 256 and ((kind == 4 and ncbs[0] > 12 * ncu) or kind in (1, 3)), _lib.last_error()
            continue
        cover = [np.zeros((c, nblk), dtype=np.int32) for c in ncbs]
        slices = {}
        ub = 0
        for mat, cb0, ncb, b0, nb, sl, side, flags, ubase, pA, pB, pC in tiles:
            if mat < 0: continue
            # the rectangle's place in the op's repacked weights, and the streaming waves' shares of it (uniform: all zero; weighted: n << 2 | e per age group)
            assert ubase == ub
            ub += 4 * nb * ncb
            _check_wave_shares(4 * nb * ncb, 4 * nb, (pA, pB, pC), head=(kind == 4))
            assert 0 <= mat < len(ncbs) and ncb >= 1 and nb >= 1 and 0 <= sl < S.value
            assert ncb <= 12
            if kind in (1, 3): assert nb <= 8 and ncb <= 4
            if kind == 4: assert S.value == 1
            cover[mat][cb0:cb0 + ncb, b0:b0 + nb] += 1
            slices.setdefault((mat, cb0), []).append(sl)
        for c in cover: assert (c == 1).all(), (name, kind, ncu)
        for k, v in slices.items(): assert sorted(v) == list(range(S.value)), (k, v)
        sides = sorted(int(t[6]) for t in tiles if t[6] >= 0)
        assert sides == (list(range(2 * kvdim // 128)) if kind == 1 else [])
        assert sum(1 for t in tiles if t[0] >= 0 and (t[7] & 1)) == (S.value if True else 0)           # one column group per slice carries the q-out flag


def _wave_runs(T, shares):
    """exl3_pstep_kernel.cuh: ps_wave_range, restated."""
    pA, pB, pC = shares
    if pA == 0 and pB == 0 and pC == 0:
        return [((T * w) // 12, (T * (w + 1)) // 12) for w in range(12)]
    out, base = [], 0
    for p in (pA, pB, pC):
        n, e = p >> 2, p & 3
        for i in range(4):
            u0 = base + i * n + min(i, e)
            out.append((u0, u0 + n + (1 if i < e else 0)))
        base += 4 * n + e
    return out


def _check_wave_shares(T, H, shares, head=False):
    """The twelve runs tile [0, T) in wave order, no run is longer than a column (it may cross ONE column boundary: a wave keeps two partial rows), older age groups take
    at least as much as younger ones, and every wave keeps the three units it decodes ahead."""
    runs = _wave_runs(T, shares)
    assert runs[0][0] == 0 and runs[-1][1] == T and all(runs[i][1] == runs[i + 1][0] for i in range(11))
    assert all(0 <= u1 - u0 <= H or H * 12 < T for u0, u1 in runs) or any(shares)
    if any(shares):
        assert T > 36                                 # (round 6: the lm_head's rectangles are cut by age group too -- its partial rows take the service-wave sum)
        n = [u1 - u0 for u0, u1 in runs]
        assert max(n) <= H and min(n) >= 3 and min(n[0:4]) >= max(n[4:8]) - 1 and min(n[4:8]) >= max(n[8:12]) - 1 and n[0] > n[11]
    for u0, u1 in runs:
        if u1 > u0: assert (u1 - 1) // H - u0 // H <= 1
    # the finish reads the partial rows of a column from at most eight consecutive waves (exl3_pstep_kernel.cuh: col_table)
    for j in range(T // H if T > H else 0):          # (a one-column rectangle: rows 0-5 | 6-11 of all twelve waves, no table)
        touch = [w for w, (u0, u1) in enumerate(runs) if u1 > u0 and u0 < (j + 1) * H and u1 > j * H]
        assert touch and touch[-1] - touch[0] < 8


def test_weighted_partition_is_what_the_bench_shapes_get():
    """Llama-3.1-8B on 256 CUs: gate|up (112 units per rectangle), down (56) and the lm_head are cut by age group, q|k|v / o (all units decoded ahead) stay uniform;
    EXL3_HIP_PSTEP_SHARES=0,0 turns the weighting off."""
    import os
    from exllamav3_amd import _lib
    l = _lib.lib()

    def plan(kind):
        tiles = np.zeros((256, 12), dtype=np.int32); S = ctypes.c_int(0)
        assert l.exl3_pstep_plan_tiles(4096, 14336, 32, 8, 128, 128256, 256, kind, tiles.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.byref(S)) == 0
        return tiles
    for kind, weighted in ((0, False), (1, False), (2, True), (3, True), (4, True)):
        t = plan(kind)
        act = t[t[:, 0] >= 0]
        assert bool((act[:, 9:12] != 0).any()) == weighted, kind
        if weighted:
            assert (act[:, 9] >> 2 >= act[:, 10] >> 2).all() and (act[:, 10] >> 2 >= act[:, 11] >> 2).all()
    os.environ["EXL3_HIP_PSTEP_SHARES"] = "0,0"
    try:
        assert not (plan(2)[:, 9:12] != 0).any()
    finally:
        del os.environ["EXL3_HIP_PSTEP_SHARES"]


def test_planner_refuses_what_the_kernel_cannot_take():
    from exllamav3_amd import _lib
    l = _lib.lib()
    tiles = np.zeros((256, 12), dtype=np.int32); S = ctypes.c_int(0)
    P = tiles.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    assert l.exl3_pstep_plan_tiles(4096, 14336, 32, 8, 128, 128256, 256, 7, P, ctypes.byref(S)) < 0          # no such op kind
    assert l.exl3_pstep_plan_tiles(4100, 14336, 32, 8, 128, 128256, 256, 0, P, ctypes.byref(S)) < 0          # hidden not a multiple of 128
    assert l.exl3_pstep_plan_tiles(4096, 14336, 32, 8, 96, 128256, 256, 0, P, ctypes.byref(S)) < 0           # head_dim


def test_partial_row_partition_formula_matches_the_streaming_waves_runs():
    """(The UNIFORM partition: what the lm_head's finish and the PS_SUM_HALFWAVES builds use; the weighted partitions are read off ps_wave_range by a scan over the twelve
    waves, checked in _check_wave_shares above.)
    The service waves of the persistent step find the partial rows of column j of a rectangle WITHOUT records (exl3_pstep_kernel.cuh: col_table): streaming wave w takes
    units [T w / 12, T (w + 1) / 12) of the rectangle's T = 4 nb W units, column-major (ps_make_seg), so the waves whose run touches column j are at most EIGHT consecutive
    ones from max(floor(12 j / W) - 1, 0) on, a run that starts in the column has its row in segment 0 and one that started in the column before in segment 1, and a run
    spans at most two columns (W <= 12).  Brute force over every rectangle the planner can make (nb <= 32, W <= 12): the formula's (wave, segment) set per column equals the
    set read off the runs themselves; for a one-column rectangle the rows are segment 0 of every wave that has a unit."""
    SW = 12
    for nb in range(1, 33):
        H = 4 * nb
        for W in range(1, 13):
            T = H * W
            runs = [((T * w) // SW, (T * (w + 1)) // SW) for w in range(SW)]
            truth = {j: set() for j in range(W)}
            for w, (u0, u1) in enumerate(runs):
                if u1 <= u0: continue
                cols = sorted({u // H for u in range(u0, u1)})
                assert len(cols) <= 2                                     # (a wave keeps two partial rows: segments 0 and 1)
                for seg, c in enumerate(cols): truth[c].add((w, seg))
            if W == 1:
                assert truth[0] == {(w, 0) for w, (u0, u1) in enumerate(runs) if u1 > u0}
                continue
            for j in range(W):
                lo_u, hi_u = j * H, j * H + H
                w_first = max((SW * lo_u) // T - 1, 0)
                assert (SW * lo_u) // T == (SW * j) // W
                got = set()
                for i in range(8):
                    w = min(w_first + i, SW - 1)
                    u0, u1 = runs[w]
                    if w_first + i < SW and u1 > u0 and u0 < hi_u and u1 > lo_u:
                        got.add((w, 0 if u0 >= lo_u else 1))
                assert got == truth[j], (nb, W, j, got, truth[j])
