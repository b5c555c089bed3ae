"""The one-shot IPC all-reduce of the tensor-parallel decode path (exl3_allreduce.hip) with TWO processes sharing the one GPU of the test box:
each rank allocates its fine-grained receive buffer, the 64-byte hipIpc handles travel over a gloo process group, every rank maps its peer and
the push/poll kernels of the two processes run concurrently.  Checked: plain sums (bit-exact: rank-order fp32 adds), the fused residual add +
per-block sums of squares against exl3_glue_resid on the summed tensor, many back-to-back calls (slot-set alternation), hipGraph replay, and
TPBackendRCCL.enable_ipc_allreduce's self-test.  (xGMI between 8 GPUs is first exercised by the driver's scaling run.)"""
import os
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from exllamav3_amd import ext
    from exllamav3_amd.tp import TPBackendRCCL
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ext.init(0)
    be = TPBackendRCCL(rank, world, dev, backend="gloo")
    ok = {}
    try:
        ok["enabled"] = be.enable_ipc_allreduce(4 * 8192)
        ipc = be.ipc
        g = torch.Generator(device=dev); g.manual_seed(100 + rank)
        # ---- plain sums, several shapes, 40 back-to-back calls (slot sets alternate; a rank may run one call ahead of its peer)
        exact = True
        for it in range(40):
            m, hidden = [(1, 4096), (4, 8192), (2, 2048), (3, 128)][it % 4]
            y = torch.randn((m, hidden), device=dev, generator=g)
            ref = y.clone(); dist.all_reduce(ref)                      # gloo: the sum of the two ranks' tensors
            out = torch.empty_like(y)
            ipc.reduce(y, y_out=out)
            exact = exact and bool(torch.equal(out, ref))
        ok["sums_exact"] = exact and ipc.error() == 0
        # ---- fused residual add + sums of squares == all_reduce + glue_resid
        m, hidden = 2, 4096
        y = torch.randn((m, hidden), device=dev, generator=g)
        r0 = (torch.randn((m, hidden), device=dev, generator=torch.Generator(device=dev).manual_seed(7)) * 2).half()   # same on both ranks
        ysum = y.clone(); dist.all_reduce(ysum)
        r_ref = r0.clone(); ss_ref = torch.zeros((m, hidden // 128), device=dev)
        ext.glue_resid(None, 0, None, None, r_ref, ss_ref, m, y_dense=ysum)
        r1 = r0.clone(); ss1 = torch.full_like(ss_ref, float("nan"))
        be.all_reduce_resid(y, r1, ss1, m)
        ok["resid_equal"] = bool(torch.equal(r1, r_ref)) and bool(torch.equal(ss1, ss_ref))
        # ---- hipGraph: three reductions captured once, replayed five times with fresh inputs
        ys = [torch.zeros((1, 4096), device=dev) for _ in range(3)]
        outs = [torch.empty_like(t) for t in ys]
        st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
        dist.barrier()
        with torch.cuda.stream(st):
            for a, b in zip(ys, outs): ipc.reduce(a, y_out=b)          # warm-up outside capture (both ranks: same call sequence)
            st.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=st):
                for a, b in zip(ys, outs): ipc.reduce(a, y_out=b)
        torch.cuda.synchronize()
        gok = True
        for rep in range(5):
            for t in ys: t.copy_(torch.randn(t.shape, device=dev, generator=g))
            refs = []
            for t in ys:
                r = t.clone(); dist.all_reduce(r); refs.append(r)
            graph.replay(); torch.cuda.synchronize()
            gok = gok and all(bool(torch.equal(o_, r_)) for o_, r_ in zip(outs, refs))
        ok["graph_replay"] = gok and ipc.error() == 0
        # ---- the all-reduce launch as the epilogue of a row-sharded linear: deferred slabs in, residual out == GEMV (fp32 out) + all_reduce_resid
        import sys as _s
        _s.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
        import exl3_oracle as o
        k, n, K, m = 1024, 2048, 4, 2
        tr, suh, svh = o.synth_linear(k, n, K, seed=500 + rank, realistic=True)            # each rank its own row shard
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        dtr, dsuh, dsvh = T(tr), T(suh), T(svh)
        xin = torch.randn((m, k), device=dev, generator=g).half()
        ydense = torch.empty((m, n), dtype=torch.float, device=dev)
        ext.exl3_gemv_ex(xin, None, None, [dtr], [ydense], [dsuh], [dsvh], m, False, True, 0, c_fp32=True)
        rA = r0[:, :n].contiguous().clone(); ssA = torch.zeros((m, n // 128), device=dev)
        be.all_reduce_resid(ydense, rA, ssA, m)
        slabs, S = ext.exl3_gemv_ex(xin, None, None, [dtr], None, [dsuh], None, m, False, True, ext.GEMV_OUT_DEFERRED)
        rB = r0[:, :n].contiguous().clone(); ssB = torch.full_like(ssA, float("nan"))
        be.all_reduce_resid_slabs(slabs[0], S, dsvh, rB, ssB, m)
        torch.cuda.synchronize()
        ok["slab_route"] = bool((rA.float() - rB.float()).abs().max() <= 2e-3 * rA.float().abs().max()) and bool(torch.allclose(ssA, ssB, rtol=2e-3)) and ipc.error() == 0
        # ---- a two-rank tensor-parallel decode step through both routes (dense partials vs slabs into the all-reduce), eager and graph replay
        from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
        shape = LlamaShape("tiny", 512, 1024, 2, 4, 2, 128, 384)
        model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, backend=be, kv_bits=4, max_ctx=1024)
        model.alloc_state(1, pos=50)
        dist.broadcast(model.x0, 0)                                      # one input row for both ranks (alloc_state draws it from the rank's own generator)
        model.ar_from_slabs = False
        la = model.decode_step_fused().float().clone()
        model.ar_from_slabs = True
        lb = model.decode_step_fused().float().clone()
        rms = float(la.pow(2).mean().sqrt())
        step_ok = bool(torch.isfinite(lb).all()) and float((la - lb).abs().max()) / rms < 1e-2
        st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
        dist.barrier()
        with torch.cuda.stream(st):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=st):
                model.decode_step_fused()
        torch.cuda.synchronize(); dist.barrier()
        for _ in range(3):
            model.logits.zero_(); graph.replay()
        torch.cuda.synchronize()
        ok["tp_step_slab_route"] = step_ok and bool(torch.equal(model.logits.float(), lb)) and ipc.error() == 0
        # ---- the fx pipeline under TP (round 4): R += rank-order sum inside the all-reduce launch; logits == the glue TP step to fp16 rounding, the
        # fixed-point residual BIT-identical on both ranks (gathered below), the library route (fp32 partials -> collective -> fx_add) agrees, graph replay
        lfx = model.decode_step_fx().float().clone()
        ok["fx_tp_close_to_glue_tp"] = bool(torch.isfinite(lfx).all()) and float((lfx - la).abs().max()) / rms < 1.5e-2
        Rmine = model.R.clone()
        Rall = [torch.empty_like(Rmine) for _ in range(world)]
        dist.all_gather(Rall, Rmine)
        ok["fx_tp_residual_bit_identical_across_ranks"] = all(bool(torch.equal(Rall[0], r)) for r in Rall)
        saved = be.ipc; be.ipc = None
        llib = model.decode_step_fx().float().clone()
        be.ipc = saved
        ok["fx_tp_library_route"] = float((llib - lfx).abs().max()) / rms < 1e-2
        st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
        dist.barrier()
        with torch.cuda.stream(st):
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=st):
                model.decode_step_fx()
        torch.cuda.synchronize(); dist.barrier()
        for _ in range(3):
            model.logits.zero_(); graph.replay()
        torch.cuda.synchronize()
        ok["fx_tp_graph_replay"] = bool(torch.equal(model.logits.float(), lfx)) and ipc.error() == 0
    except Exception as e:           # report instead of hanging the peer
        ok["exception"] = repr(e)
    ret[rank] = ok
    try:
        be.close()
    except Exception:
        pass


def test_ipc_allreduce_two_processes_one_gpu(dev):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29800 + (os.getpid() % 150)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        res = ret.get(r)
        assert res and "exception" not in res, res
        assert res == {"enabled": True, "sums_exact": True, "resid_equal": True, "graph_replay": True, "slab_route": True, "tp_step_slab_route": True,
                       "fx_tp_close_to_glue_tp": True, "fx_tp_residual_bit_identical_across_ranks": True, "fx_tp_library_route": True,
                       "fx_tp_graph_replay": True}, (r, res)


def _worker8(rank, world, port, ret):
    """EIGHT ranks on the one GPU of the test box (the collectives over gloo, the all-reduce over the IPC push): what the driver's 8-GPU run does first
    on real links -- rank-order sums bit-identical on every rank, the fx tensor-parallel step with a bit-identical residual on all eight, and the
    time-out -> NaN -> every-rank-falls-back path for a rank that stops taking part."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from exllamav3_amd import ext
    from exllamav3_amd.tp import TPBackendRCCL
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ext.init(0)
    be = TPBackendRCCL(rank, world, dev, backend="gloo")
    ok = {}
    try:
        ok["enabled"] = be.enable_ipc_allreduce(2 * 4096)
        ipc = be.ipc
        g = torch.Generator(device=dev); g.manual_seed(300 + rank)
        exact = True
        for it in range(12):
            m, hidden = [(1, 4096), (2, 1024), (1, 128)][it % 3]
            y = torch.randn((m, hidden), device=dev, generator=g)
            parts = [torch.empty_like(y) for _ in range(world)]
            dist.all_gather(parts, y)
            ref = parts[0].clone()
            for r in range(1, world): ref += parts[r]                    # the rank-order fp32 sum the kernel computes: the same bits on every rank
            out = torch.empty_like(y)
            ipc.reduce(y, y_out=out)
            exact = exact and bool(torch.equal(out, ref))
        ok["rank_order_sums_bit_exact"] = exact and ipc.error() == 0
        # ---- the fx pipeline under TP = 8: the fixed-point residual bit-identical on all ranks
        from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
        shape = LlamaShape("tiny8", 1024, 1024, 2, 8, 8, 128, 1024)
        model = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, backend=be, kv_bits=4, max_ctx=1024)
        model.alloc_state(1, pos=50)
        dist.broadcast(model.x0, 0)
        lfx = model.decode_step_fx().float().clone()
        Rall = [torch.empty_like(model.R) for _ in range(world)]
        dist.all_gather(Rall, model.R.clone())
        ok["fx_tp8_residual_bit_identical"] = bool(torch.isfinite(lfx).all()) and all(bool(torch.equal(Rall[0], r)) for r in Rall) and ipc.error() == 0
        ok["still_on"] = be.poll_ipc_allreduce()
        # ---- a rank that stops taking part: the others' bounded spins give up, poison their outputs with NaN and raise the error word; the collective
        # health check then takes EVERY rank off the IPC path together and later reductions go through the collective library
        y = torch.ones((1, 1024), device=dev) * (rank + 1)
        out = torch.zeros_like(y)
        if rank != world - 1:
            ipc.reduce(y, y_out=out)
            torch.cuda.synchronize()
            ok["timeout_poisons_with_nan"] = bool(torch.isnan(out).any())
        else:
            ok["timeout_poisons_with_nan"] = True                        # (this rank sat the call out)
        ok["every_rank_falls_back"] = (be.poll_ipc_allreduce() is False) and be.ipc is None
        r0 = torch.zeros((1, 1024), dtype=torch.half, device=dev); ss = torch.zeros((1, 8), device=dev)
        be.all_reduce_resid(y.clone(), r0, ss, 1)                        # collective-library route
        torch.cuda.synchronize()
        ok["library_route_after_fallback"] = bool(torch.equal(r0.float(), torch.full_like(r0.float(), float(sum(range(1, world + 1))))))
    except Exception as e:
        ok["exception"] = repr(e)
    ret[rank] = ok
    try:
        be.close()
    except Exception:
        pass


def test_ipc_allreduce_eight_processes_one_gpu(dev):
    world = 8
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29960 + (os.getpid() % 30)
    mp.spawn(_worker8, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        res = ret.get(r)
        assert res and "exception" not in res, (r, res)
        assert res == {"enabled": True, "rank_order_sums_bit_exact": True, "fx_tp8_residual_bit_identical": True, "still_on": True,
                       "timeout_poisons_with_nan": True, "every_rank_falls_back": True, "library_route_after_fallback": True}, (r, res)
