"""Tensor-parallel prefill parity (task 3) with TWO processes sharing the one GPU of the test box (gloo collectives): a TP = 2 model
whose ranks read their column / row shards of ONE checkpoint (loader tp_slice) runs prefill_chunk -- q/k/v/gate/up column shards, o/down row shards with
fp32 partial sums all-reduced, rms_norm_res_in on the reduced sum -- and must reproduce the TP = 1 model of the same checkpoint on the same chunk:
residual stream on sampled token rows and the last token's logits (gathered from the lm_head column shards).  Also exercised here: the collective
health poll of the IPC decode all-reduce (tp.poll_ipc_allreduce: error words + epoch lockstep) and the collective fallback to the library path."""
import os
import tempfile
import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, ckpt, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from exllamav3_amd import ext
    from exllamav3_amd.tp import TPBackendRCCL
    from exllamav3_amd.llama_path import SyntheticEXL3Llama
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ext.init(0)
    be = TPBackendRCCL(rank, world, dev, backend="gloo")
    ok = {}
    try:
        tokens = 256
        tp = SyntheticEXL3Llama.from_checkpoint(ckpt, device=dev, backend=be, kv_bits=4, max_ctx=1024)
        assert tp.tp == 2
        tp.prefill_chunk(tokens)                                         # creates the chunk buffers (random input)
        g = torch.Generator(device=dev); g.manual_seed(4242)             # the same chunk on every rank and for the TP = 1 model
        x_common = torch.randn((tokens, tp.shape.hidden), device=dev, generator=g).half()
        tp.px0.copy_(x_common)
        lg_local = tp.prefill_chunk(tokens).float()                      # (1, vocab / 2): this rank's lm_head column shard
        parts = [torch.empty_like(lg_local) for _ in range(world)]
        dist.all_gather(parts, lg_local)
        lg_tp = torch.cat(parts, dim=-1)
        x_tp = tp.px_out.float().clone()
        ok["finite"] = bool(torch.isfinite(lg_tp).all()) and bool(torch.isfinite(x_tp).all())
        # both ranks hold the same residual stream after the all-reduces
        xs = [torch.empty_like(x_tp) for _ in range(world)]
        dist.all_gather(xs, x_tp)
        ok["ranks_agree"] = bool(torch.equal(xs[0], xs[1]))
        if rank == 0:
            one = SyntheticEXL3Llama.from_checkpoint(ckpt, device=dev, backend=None, kv_bits=4, max_ctx=1024)
            assert one.tp == 1
            one.prefill_chunk(tokens); one.px0.copy_(x_common)
            lg_one = one.prefill_chunk(tokens).float()
            x_one = one.px_out.float()
            rows = torch.tensor([0, 1, 37, 128, 255], device=dev)
            rx = float((x_tp[rows] - x_one[rows]).abs().max()) / float(x_one[rows].pow(2).mean().sqrt())
            rl = float((lg_tp - lg_one).abs().max()) / float(lg_one.pow(2).mean().sqrt())
            ok["resid_rel"] = rx; ok["logit_rel"] = rl
            ok["parity"] = rx < 1e-2 and rl < 2e-2
        dist.barrier()
        # ---- collective health poll / fallback of the IPC decode all-reduce
        enabled = be.enable_ipc_allreduce(4 * 4096)
        ok["ipc_enabled"] = enabled
        if enabled:
            y = torch.randn((1, 512), device=dev); r = torch.zeros((1, 512), dtype=torch.half, device=dev); ss = torch.zeros((1, 4), device=dev)
            for _ in range(5):
                be.all_reduce_resid(y, r, ss, 1)
            ok["poll_true"] = be.poll_ipc_allreduce() is True
            be.disable_ipc_allreduce()
            ok["disabled"] = be.ipc is None and be.poll_ipc_allreduce() is False
            ysum = y.clone(); dist.all_reduce(ysum)
            r2 = torch.zeros((1, 512), dtype=torch.half, device=dev)
            be.all_reduce_resid(y.clone(), r2, ss, 1)                       # library path now
            ok["fallback_correct"] = bool(torch.allclose(r2.float(), ysum, rtol=2e-3, atol=2e-3))
    except Exception as e:
        import traceback
        ok["exception"] = repr(e) + traceback.format_exc()[-1500:]
    ret[rank] = ok
    try:
        be.close()
    except Exception:
        pass


def test_tp2_prefill_chunk_matches_tp1_on_one_gpu(dev):
    from exllamav3_amd.llama_path import LlamaShape, SyntheticEXL3Llama
    shape = LlamaShape("tiny", 512, 1024, 2, 4, 2, 128, 512)             # vocab: two equal lm_head column shards
    with tempfile.TemporaryDirectory() as td:
        SyntheticEXL3Llama(shape, K=4, cb=2, device=dev, kv_bits=4, max_ctx=1024).save_checkpoint(td)
        world = 2
        mgr = mp.Manager()
        ret = mgr.dict()
        port = 29400 + (os.getpid() % 150)
        mp.spawn(_worker, args=(world, port, td, ret), nprocs=world, join=True)
    for r in range(world):
        res = ret.get(r)
        assert res and "exception" not in res, res
        assert res["finite"] and res["ranks_agree"], (r, res)
        assert res["ipc_enabled"] and res["poll_true"] and res["disabled"] and res["fallback_correct"], (r, res)
    assert ret[0]["parity"], ret[0]
