import os, sys, numpy as np, torch, torch.multiprocessing as mp
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
def worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from test_gpu_pstep_tp import _masked_stream
    from test_gpu_pstep import _kv_levels4
    dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
    stream, per = _masked_stream(rank, world, 256)
    os.environ["EXL3_HIP_PSTEP_NCU"] = str(per)
    from exllamav3_amd import ext
    from exllamav3_amd.tp import TPBackendRCCL
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    ext.init(0); ext.set_gemv_variant(1)
    be = TPBackendRCCL(rank, world, dev, backend="gloo")
    shape = LlamaShape("tp-ps", 1024, 2816, 2, 8, 2, 128, 3072)
    m = SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, backend=be, kv_bits=4, max_ctx=1024)
    m.alloc_state(1, pos=300)
    dist.broadcast(m.x0, 0)
    be.enable_ipc_allreduce(1024)
    lf = m.decode_step_fx().float().cpu().numpy().copy()
    kv_f = [(c.cpu().numpy().copy(), s_.cpu().numpy().copy()) for c, s_ in m.kcache + m.vcache]
    qf = m.q.float().cpu().numpy().copy()
    for c, s_ in m.kcache + m.vcache: c.zero_(); s_.zero_()
    torch.cuda.synchronize(); dist.barrier()
    out = {}
    with torch.cuda.stream(stream):
        lp = m.decode_step_persistent().float().cpu().numpy().copy()
        stream.synchronize()
        out["plan"] = m._pstep.describe() if m._pstep else None
        out["err"] = m._pstep.error() if m._pstep else None
        rms = float(np.sqrt((lf ** 2).mean()))
        out["logits"] = float(np.abs(lp - lf).max() / rms)
        out["q"] = float(np.abs(m.q.float().cpu().numpy() - qf).max())
        page, slot = 300 // 256, 300 % 256
        for i, ((wa, sa), (c, s_)) in enumerate(zip(kv_f, m.kcache + m.vcache)):
            la, lb = _kv_levels4(wa[page, slot]), _kv_levels4(c[page, slot].cpu().numpy())
            out["kv%d" % i] = (int(np.abs(la - lb).max()), float((la != lb).mean()))
    import ctypes
    from exllamav3_amd import _lib
    buf = (ctypes.c_uint32 * (1 << 22))()
    n = _lib.lib().exl3_pstep_tp_peek(m._pstep._h, buf, 4 << 22)
    xb = np.frombuffer(buf, dtype=np.uint32, count=n // 4).copy()
    t = torch.from_numpy(xb.astype(np.int64))
    both = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(both, t)
    a0, a1 = both[0].numpy(), both[1].numpy()
    out["xbuf_bytes"] = int(n); out["xbuf_equal_words"] = float((a0 == a1).mean())
    # first lines: tags (word 1 of each 16-byte granule) per 512-byte line
    lines = xb.reshape(-1, 128)          # 512 B = 128 words
    tags = lines[:, 1]
    nz = np.nonzero(tags)[0]
    out["lines_with_tags"] = int(len(nz)); out["first_tags"] = [hex(int(v)) for v in tags[:24]]
    d = np.nonzero(a0 != a1)[0]
    out["first_diff_words"] = [int(v) for v in d[:8]]
    ret[rank] = out
    be.close()
if __name__ == "__main__":
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(worker, args=(2, 29991, ret), nprocs=2, join=True)
    for r in range(2): print(r, ret.get(r))
