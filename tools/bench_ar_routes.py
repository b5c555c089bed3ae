#!/usr/bin/env python3
"""Two processes on ONE GPU (gloo for the handles, IPC for the data): a row-sharded o_proj-sized linear + the tensor-parallel boundary,
(a) GEMV with its own fp32 epilogue + all_reduce_resid (dense partial) vs (b) deferred GEMV + all_reduce_resid_slabs; hipGraph of 32 boundaries,
per-boundary microseconds.  Also the two launches of each route on their own."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch, torch.multiprocessing as mp


def worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from exllamav3_amd import ext
    from exllamav3_amd.tp import TPBackendRCCL
    import exl3_oracle as o
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev); ext.init(0)
    be = TPBackendRCCL(rank, world, dev, backend="gloo")
    assert be.enable_ipc_allreduce(4 * 8192)
    k, n, K, m, reps = 2048, 4096, 4, 1, 32
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    mats = []
    for i in range(reps):                                              # distinct weights per boundary: cold like a real step
        tr, suh, svh = o.synth_linear(k, n, K, seed=i if i < 4 else i % 4, realistic=False)
        mats.append((T(tr).clone(), T(suh), T(svh)))
    x = torch.randn((m, k), device=dev).half()
    resid = torch.zeros((m, n), dtype=torch.half, device=dev); ss = torch.zeros((m, n // 128), device=dev)
    yd = torch.empty((m, n), dtype=torch.float, device=dev)

    def route_dense():
        for tr, suh, svh in mats:
            ext.exl3_gemv_ex(x, None, None, [tr], [yd], [suh], [svh], m, False, True, 0, c_fp32=True)
            be.all_reduce_resid(yd, resid, ss, m)

    def route_slabs():
        for tr, suh, svh in mats:
            sl, S = ext.exl3_gemv_ex(x, None, None, [tr], None, [suh], None, m, False, True, ext.GEMV_OUT_DEFERRED)
            be.all_reduce_resid_slabs(sl[0], S, svh, resid, ss, m)

    def only(fn_name):
        def f():
            for tr, suh, svh in mats:
                if fn_name == "gemv_dense": ext.exl3_gemv_ex(x, None, None, [tr], [yd], [suh], [svh], m, False, True, 0, c_fp32=True)
                elif fn_name == "gemv_deferred": ext.exl3_gemv_ex(x, None, None, [tr], None, [suh], None, m, False, True, ext.GEMV_OUT_DEFERRED)
                elif fn_name == "ar_dense": be.all_reduce_resid(yd, resid, ss, m)
        return f

    out = {}
    for name, fn in (("dense", route_dense), ("slabs", route_slabs), ("gemv_dense", only("gemv_dense")), ("gemv_deferred", only("gemv_deferred")), ("ar_dense", only("ar_dense"))):
        fn(); torch.cuda.synchronize(); dist.barrier()
        st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                fn()
        torch.cuda.synchronize(); dist.barrier()
        best = 1e9
        for rep in range(3):
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): g.replay()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 10 / reps)
        out[name] = round(best, 2)
    out["ipc_error"] = be.ipc.error()
    ret[rank] = out
    be.close()


if __name__ == "__main__":
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(worker, args=(2, 29900 + os.getpid() % 90, ret), nprocs=2, join=True)
    print(json.dumps({str(r): ret.get(r) for r in range(2)}))
