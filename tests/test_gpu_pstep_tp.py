"""Tensor parallelism INSIDE the persistent decode step (round 6; VERDICT r5 task 5): the all-reduce behind o_proj / down_proj (model/model_tp_backend.py:119-126,
modules/attn.py:547, modules/mlp.py:770) happens on the step's row edges -- every rank pushes the partial rows of its row shards as tagged lines into EVERY rank's exchange
buffer over IPC-mapped addresses, the consumers of all ranks sum the same (rank, slice) lines in the same order.

TWO processes share the one GPU of the test box.  The step needs its whole grid co-resident, so each rank launches on a stream whose CU mask holds HALF the chip
(hipExtStreamCreateWithCUMask; the plan is made for that many CUs: EXL3_HIP_PSTEP_NCU) -- both grids run at the same time and exchange lines as two GPUs would, over the same
code path except for the links.  Checked: logits of every rank's head shard against the launch-per-op TP branch (decode_step_fx under TP: IPC all-reduce launches), the
appended K / V rows level by level, both ranks' steps repeatable bit for bit, no time-out; and the collective fall-back when one rank cannot make a plan."""
import os
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _masked_stream(rank, world, ncu_total):
    """A HIP stream restricted to this rank's share of the CUs, wrapped for torch."""
    import ctypes
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    words = (ncu_total + 31) // 32
    per = ncu_total // world
    mask = (ctypes.c_uint32 * words)()
    for cu in range(rank * per, (rank + 1) * per):
        mask[cu // 32] |= 1 << (cu % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    assert rc == 0, f"hipExtStreamCreateWithCUMask: {rc}"
    return torch.cuda.ExternalStream(st.value), per


def _worker(rank, world, port, ret, case):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    ok = {}
    be = None
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        ncu_total = torch.cuda.get_device_properties(dev).multi_processor_count
        stream, per = _masked_stream(rank, world, ncu_total)
        os.environ["EXL3_HIP_PSTEP_NCU"] = str(per)
        from exllamav3_amd import ext
        from exllamav3_amd.tp import TPBackendRCCL
        from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from test_gpu_pstep import _kv_rows_match_in_levels
        ext.init(0)
        ext.set_gemv_variant(1)
        be = TPBackendRCCL(rank, world, dev, backend="gloo")
        hidden, inter, layers, hq, hkv, hd, vocab, K, att = case
        shape = LlamaShape("tp-ps", hidden, inter, layers, hq, hkv, hd, vocab)
        m = SyntheticEXL3Llama(shape, K=K, cb=2, device=dev, backend=be, kv_bits=4, max_ctx=1024)
        m.alloc_state(1, pos=300)
        m.with_attention = att
        dist.broadcast(m.x0, 0)                                          # one input row for all ranks
        if att:
            g = torch.Generator(device="cpu").manual_seed(77 + rank)
            for c, s_ in m.kcache + m.vcache:
                c.copy_(torch.randint(-2 ** 31, 2 ** 31 - 1, c.shape, generator=g, dtype=torch.int64).to(torch.int32).to(dev))
                s_.copy_((torch.rand(s_.shape, generator=g) * 0.5 + 0.05).half().to(dev))
        saved = [(c.clone(), s_.clone()) for c, s_ in m.kcache + m.vcache]

        def restore():
            for (c, s_), (c0, s0) in zip(m.kcache + m.vcache, saved): c.copy_(c0); s_.copy_(s0)
        _np = lambda t: t.detach().float().cpu().numpy()
        # ---- the launch-per-op TP branch (IPC all-reduce launches between the row shards and the norms): the reference of this test
        ok["ipc_allreduce"] = be.enable_ipc_allreduce(hidden)
        lf = _np(m.decode_step_fx()).copy()
        kv_f = [(c.cpu().numpy().copy(), s_.cpu().numpy().copy()) for c, s_ in m.kcache + m.vcache]
        restore()
        torch.cuda.synchronize(); dist.barrier()
        # ---- the persistent step, every rank on its own half of the chip
        ok["applies"] = m.persistent_applies()
        with torch.cuda.stream(stream):
            lp = _np(m.decode_step_persistent()).copy()
            stream.synchronize()
            ok["took_the_persistent_step"] = m._pstep is not None and m._pstep.tp == (world, rank)
            ok["no_timeout"] = m._pstep is not None and not m._pstep.error()
            rms = float(np.sqrt((lf.astype(np.float64) ** 2).mean()))
            ok["logits_rel_err"] = float(np.abs(lp - lf).max() / rms)
            page, slot = int(m.block_table[0, 300 // m.page]), 300 % m.page
            try:
                for (wa, sa), (c, s_) in zip(kv_f, m.kcache + m.vcache):
                    _kv_rows_match_in_levels(wa[page, slot], sa[page, slot], c[page, slot].cpu().numpy(), s_[page, slot].cpu().numpy(), 0.02)
                ok["kv_rows"] = True
            except AssertionError as e:
                ok["kv_rows"] = repr(e)
            # the same step again, and again: the same bits (rank-order sums: nothing depends on which rank's lines arrive first)
            same = True
            for _ in range(3):
                restore(); dist.barrier()
                same = same and bool(np.array_equal(_np(m.decode_step_persistent()), lp))
                stream.synchronize()
            ok["repeatable"] = same and m._pstep is not None and not m._pstep.error()
            # every rank holds the same lines: the exchange buffers of the ranks are equal word for word (what each rank's consumers sum, in the same order -> the same
            # residual rows on every rank: the bit-identity the all-reduce launches get from their rank-order sums)
            xb = torch.from_numpy(m._pstep.tp_peek().astype(np.int64))
            allx = [torch.empty_like(xb) for _ in range(world)]
            dist.all_gather(allx, xb)
            ok["exchange_buffers_identical"] = all(bool(torch.equal(allx[0], t)) for t in allx) and int((xb != 0).sum()) > 1000
        ok["plan"] = m._pstep.describe() if m._pstep is not None else None
    except Exception as e:           # report instead of hanging the peer
        import traceback
        ok["exception"] = repr(e) + traceback.format_exc()[-1500:]
    ret[rank] = ok
    try:
        if be is not None: be.close()
    except Exception:
        pass


@pytest.mark.parametrize("case", [(1024, 2816, 2, 8, 2, 128, 3072, 4, False),          # two kv heads: one per rank
                                  (2048, 4096, 3, 16, 4, 128, 2048, 3, False),         # 3 bpw, three layers (the row-edge lines alternate between two sets by layer)
                                  (1024, 2816, 2, 8, 2, 128, 3072, 4, True)])          # the decode attention inside the step, every rank over its own kv heads
def test_persistent_step_two_tensor_parallel_ranks_on_one_gpu(dev, case):
    _run_ranks(2, case)


def test_persistent_step_four_tensor_parallel_ranks_on_one_gpu(dev):
    """FOUR ranks on a quarter of the chip each (64 CUs): every row edge gathers 4 x S lines pushed by four processes -- the fan-in of the driver's first multi-GPU run."""
    _run_ranks(4, (1024, 4096, 2, 16, 4, 128, 4096, 4, False))


def _run_ranks(world, case):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29950 + (os.getpid() % 40)
    mp.spawn(_worker, args=(world, port, ret, case), nprocs=world, join=True)
    for r in range(world):
        res = ret.get(r)
        assert res and "exception" not in res, res
        assert res["applies"] and res["took_the_persistent_step"] and res["no_timeout"], res
        assert res["logits_rel_err"] < 2e-2 and res["kv_rows"] is True and res["repeatable"] and res["exchange_buffers_identical"], res


def _worker_fallback(rank, world, port, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import warnings
    import torch.distributed as dist
    ok = {}
    be = None
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        from exllamav3_amd import ext
        from exllamav3_amd.tp import TPBackendRCCL
        from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
        ext.init(0)
        be = TPBackendRCCL(rank, world, dev, backend="gloo")
        m = SyntheticEXL3Llama(LlamaShape("tp-fb", 1024, 2816, 2, 8, 2, 128, 3072), K=4, cb=2, device=dev, backend=be, kv_bits=4, max_ctx=1024)
        m.alloc_state(1, pos=300)
        dist.broadcast(m.x0, 0)
        ok["ipc_allreduce"] = be.enable_ipc_allreduce(1024)
        ref = m.decode_step_fx().float().cpu().numpy().copy()
        for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
        if rank == 1:
            def refuse(*a, **k): raise RuntimeError("no plan on this rank (test)")
            ext.PersistentStep = refuse
        torch.cuda.synchronize(); dist.barrier()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            got = m.decode_step_persistent().float().cpu().numpy().copy()
        ok["fell_back"] = m._pstep is None and any("no persistent decode step" in str(x.message) for x in w)
        ok["same_as_launch_per_op"] = bool(np.array_equal(got, ref))
        for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
        ok["stays_fallen_back"] = bool(np.array_equal(m.decode_step_persistent().float().cpu().numpy(), ref)) and m._pstep is None
    except Exception as e:
        import traceback
        ok["exception"] = repr(e) + traceback.format_exc()[-1500:]
    ret[rank] = ok
    try:
        if be is not None: be.close()
    except Exception:
        pass


def test_one_rank_without_a_plan_takes_every_rank_to_the_launch_per_op_branch(dev):
    """The set-up of a tensor-parallel persistent step is a collective: a rank that cannot make its plan says so, and EVERY rank (the one with a good plan included) runs the
    launch-per-op TP branch -- the same bits as decode_step_fx -- now and on later calls."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29900 + (os.getpid() % 40)
    mp.spawn(_worker_fallback, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        res = ret.get(r)
        assert res and "exception" not in res, res
        assert res["fell_back"] and res["same_as_launch_per_op"] and res["stays_fallen_back"], res
