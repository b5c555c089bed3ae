"""CPU, world_size = 2 over gloo: the tensor-parallel host logic (split points, LinearEXL3.tp_shard slicing semantics of
modules/quant/exl3.py:284-330, TPBackendRCCL all_reduce / gather / broadcast) reproduces the unsharded result.
The per-rank linear is evaluated with the oracle (no GPU in this container); the N > 1 GPU path runs the same host
code with the HIP kernels underneath."""
import os
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _worker(rank, world, port, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from oracle import exl3_oracle as o
    from exllamav3_amd.tp import TPBackendRCCL, split_points
    from exllamav3_amd.linear import LinearEXL3
    be = TPBackendRCCL(rank, world, torch.device("cpu"), backend="gloo")
    k, n, inter, K, cb = 256, 384, 640, 4, 2
    x = np.random.default_rng(0).standard_normal((2, k)).astype(np.float16)
    up = o.synth_linear(k, inter, K, seed=1, realistic=True)
    down = o.synth_linear(inter, n, K, seed=2, realistic=True)
    bias = (np.random.default_rng(3).standard_normal(n) * 0.1).astype(np.float16)
    T = torch.from_numpy
    lin_up = LinearEXL3(k, inter, T(up[0]), T(up[1]), T(up[2]), mul1=True)
    lin_down = LinearEXL3(inter, n, T(down[0]), T(down[1]), T(down[2]), mul1=True, bias=T(bias), out_dtype=torch.float)
    pts = split_points(inter, world)
    su = lin_up.tp_shard(pts[rank], pts[rank + 1], "n")          # column shard
    sd = lin_down.tp_shard(pts[rank], pts[rank + 1], "k")        # row shard
    ev = lambda L, a, f32: o.linear_forward(a, L.trellis.numpy(), L.suh.numpy(), L.svh.numpy(), L.K, cb,
                                            bias=None if L.bias is None else L.bias.numpy(), out_fp32=f32)
    h = ev(su, x, False)
    part = torch.from_numpy(ev(sd, h, True).copy())
    be.all_reduce(part)
    # unsharded reference
    h_full = ev(lin_up, x, False)
    y_full = ev(lin_down, h_full, True)
    ok1 = np.allclose(h, h_full[:, pts[rank]:pts[rank + 1]], atol=1e-3, rtol=1e-3)
    ok2 = np.abs(part.numpy() - y_full).max() / np.sqrt((y_full ** 2).mean()) < 5e-3
    # gather of column shards (lm_head logits), uneven widths
    out = torch.zeros((2, inter), dtype=torch.half) if rank == 0 else None
    be.gather(torch.from_numpy(h.copy()), out, 0, [pts[r + 1] - pts[r] for r in range(world)])
    ok3 = True if rank != 0 else np.array_equal(out.numpy(), h_full)
    # broadcast + MAX reduce (bench.py's timing reduction)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    be.all_reduce_max(t)
    b = torch.full((4,), float(rank))
    be.broadcast(b, 1)
    ok4 = float(t) == world and bool((b == 1).all())
    ret[rank] = bool(ok1 and ok2 and ok3 and ok4)
    be.close()


def test_tp2_shards_and_collectives_gloo():
    from exllamav3_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29600 + (os.getpid() % 200)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_split_points():
    from exllamav3_amd.tp import split_points
    assert split_points(14336, 8) == [0, 1792, 3584, 5376, 7168, 8960, 10752, 12544, 14336]
    p = split_points(128256, 8)
    assert p[0] == 0 and p[-1] == 128256 and all((b - a) % 128 == 0 and b > a for a, b in zip(p, p[1:]))
    assert split_points(640, 2) == [0, 384, 640]
