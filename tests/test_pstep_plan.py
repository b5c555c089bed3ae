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
        tiles = np.zeros((ncu, 8), dtype=np.int32); S = ctypes.c_int(0)
        rc = l.exl3_pstep_plan_tiles(hidden, inter, hq, hkv, hd, vocab, ncu, kind, tiles.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), ctypes.byref(S))
        if rc != 0:
            # legitimate refusals, small chips only: an lm_head wider than 12 column blocks per CU (a streaming wave's run would cross two boundaries), or an
            # op that adds into the residual row whose <= 8-block slices leave column groups wider than the 4 blocks an owner gathers
            assert ncu < 256 and ((kind == 4 and ncbs[0] > 12 * ncu) or kind in (1, 3)), _lib.last_error()
            continue
        cover = [np.zeros((c, nblk), dtype=np.int32) for c in ncbs]
        slices = {}
        for mat, cb0, ncb, b0, nb, sl, side, flags in tiles:
            if mat < 0: continue
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


def test_planner_refuses_what_the_kernel_cannot_take():
    from exllamav3_amd import _lib
    l = _lib.lib()
    tiles = np.zeros((256, 8), dtype=np.int32); S = ctypes.c_int(0)
    P = tiles.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
    assert l.exl3_pstep_plan_tiles(4096, 14336, 32, 8, 128, 128256, 256, 7, P, ctypes.byref(S)) < 0          # no such op kind
    assert l.exl3_pstep_plan_tiles(4100, 14336, 32, 8, 128, 128256, 256, 0, P, ctypes.byref(S)) < 0          # hidden not a multiple of 128
    assert l.exl3_pstep_plan_tiles(4096, 14336, 32, 8, 96, 128256, 256, 0, P, ctypes.byref(S)) < 0           # head_dim
