"""
Tensor-parallel shards of the quantized Linear + the collective backend, one process per GPU.

Mirrors the reference's seam `TPBackendNCCL` (/root/reference/exllamav3/model/model_tp_backend.py:38-180): the same
method names (fwd_barrier, broadcast, all_reduce, gather).  On ROCm `torch.distributed` backend "nccl" IS RCCL over
xGMI; on CPU (tests) the same code runs over "gloo".  Unlike the reference, broadcast and gather also go through the
collective library instead of a pinned-host-memory side channel (the reference's fallback was designed for PCIe
consumer GPUs without P2P, model_tp_backend.py:113-114,137).

Partitioning (reference modules/quant/exl3.py:284-330, modules/mlp.py:833-891, modules/attn.py:915-960):
    q / k / v / gate / up : column (out-feature) shards      -> no exchange
    o / down              : row (in-feature) shards          -> ONE all-reduce(sum) of (tokens, hidden) each
    lm_head               : column shards + gather of logits
Split points are multiples of 128 (Hadamard blocks); attention splits whole KV-head groups.
"""
from __future__ import annotations
import ctypes
import os
import torch
import torch.distributed as dist


def split_points(total: int, parts: int, unit: int = 128) -> list[int]:
    """Boundaries [0, ..., total] of `parts` contiguous ranges, each a multiple of `unit` (as even as possible)."""
    assert total % unit == 0
    units = total // unit
    base, rem = divmod(units, parts)
    pts = [0]
    for r in range(parts):
        pts.append(pts[-1] + (base + (1 if r < rem else 0)) * unit)
    return pts


class TPBackendRCCL:
    """all_reduce / barrier / broadcast / gather over torch.distributed (RCCL on GPU, gloo on CPU)."""

    def __init__(self, rank: int | None = None, world_size: int | None = None, device: torch.device | None = None,
                 backend: str | None = None, bf16_wire: bool = False):
        self.rank = int(os.environ.get("RANK", 0)) if rank is None else rank
        self.world_size = int(os.environ.get("WORLD_SIZE", 1)) if world_size is None else world_size
        self.device = device
        # The reference halves the wire by casting fp32 -> bf16 -> fp32 (model_tp_backend.py:120-124); that changes
        # numerics, so it is opt-in here.
        self.bf16_wire = bf16_wire
        self.ipc = None
        if self.world_size > 1 and not dist.is_initialized():
            if backend is None:
                backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world_size, **kw)

    def close(self):
        if self.ipc is not None:
            self.ipc.close()
            self.ipc = None
        if self.world_size > 1 and dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()

    # ---- decode all-reduce over IPC-mapped peer buffers (exl3_allreduce.hip): one hop over xGMI, fused with the residual add ----------
    def enable_ipc_allreduce(self, max_elems: int, self_test: bool = True) -> bool:
        """Set up the one-shot push all-reduce for messages of up to max_elems fp32 values (tokens x hidden of a decode step).  The 64-byte IPC
        handles travel through the process group; with self_test the first reduction is checked against the collective library and the path
        stays off (returns False, RCCL keeps doing the decode all-reduces) if a peer cannot be mapped or the sums differ."""
        if self.world_size == 1 or self.device is None or self.device.type != "cuda":
            return False
        # every rank issues the SAME sequence of collectives whatever fails locally (a rank that skipped one would pair its next collective
        # with a different one of its peers): local steps are wrapped, agreement steps are unconditional
        def agree(ok: bool) -> bool:
            flag = torch.tensor([1.0 if ok else 0.0], device=self.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return float(flag.item()) == 1.0

        ipc = IpcAllReduce.create(self.rank, self.world_size, self.device, max_elems, agree)
        if ipc is None:
            return False
        if self_test:
            g = torch.Generator(device=self.device); g.manual_seed(1234 + self.rank)
            y = torch.randn((1, 256), device=self.device, generator=g)
            ref = y.clone(); dist.all_reduce(ref)
            ok = True
            try:
                out = torch.empty_like(y)
                ipc.reduce(y, y_out=out)
                ok = ipc.error() == 0 and bool(torch.allclose(out, ref, rtol=1e-5, atol=1e-5))
            except Exception:
                ok = False
            if not agree(ok):
                torch.cuda.synchronize(self.device)
                dist.barrier()                                              # nobody is still pushing into a buffer that is about to go away
                ipc.close()
                dist.barrier()
                return False
        self.ipc = ipc
        return True

    def _agree(self, ok: bool) -> bool:
        flag = torch.tensor([1.0 if ok else 0.0], device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return float(flag.item()) == 1.0

    def poll_ipc_allreduce(self) -> bool:
        """Collective health check of the IPC all-reduce, to be called by EVERY rank at the same point of the decode loop (bench.py: after warm-up
        and after the timed region; a generation loop: every few dozen steps -- it synchronises the stream).  Reads this rank's error word (a bounded
        spin gave up: the affected elements are NaN) and epoch counter, agrees over the process group, and if any rank saw a timeout or the epochs
        are out of lockstep EVERY rank drops the path together: barrier (no peer is still pushing), unmap, free; later all_reduce_resid calls go
        through the collective library.  Returns True while the IPC path is on.  A caller that captured hipGraphs with the IPC launches inside must
        re-capture after a False."""
        if self.ipc is None:
            return False
        ok = True
        epoch = -1
        try:
            ok = self.ipc.error() == 0
            epoch = self.ipc.epoch()
        except Exception:
            ok = False
        ep = torch.tensor([float(epoch), -float(epoch)], dtype=torch.float64, device=self.device)
        dist.all_reduce(ep, op=dist.ReduceOp.MAX)                            # max(epoch) and -min(epoch)
        ok = ok and float(ep[0].item()) == -float(ep[1].item())
        if self._agree(ok):
            return True
        self.disable_ipc_allreduce()
        return False

    def disable_ipc_allreduce(self):
        """Collective: every rank stops using the IPC path.  The barrier comes first so that no peer is still pushing into a buffer that is about to be
        unmapped or freed."""
        if self.ipc is None:
            return
        torch.cuda.synchronize(self.device)
        dist.barrier()
        self.ipc.close()
        self.ipc = None
        dist.barrier()

    def all_reduce_resid(self, y: torch.Tensor, resid: torch.Tensor, ss_part: torch.Tensor, m: int):
        """resid (fp16) += sum over ranks of y (fp32 [m][hidden]); ss_part = per-block sums of squares of the new residual: the o_proj / down_proj
        boundary of a tensor-parallel decode step.  One kernel on the IPC path; all_reduce + glue_resid otherwise."""
        if self.ipc is not None and y.numel() <= self.ipc.max_elems:
            self.ipc.reduce(y, resid=resid, ss_part=ss_part, m=m)
            return
        from . import ext
        self.all_reduce(y)
        ext.glue_resid(None, 0, None, None, resid, ss_part, m, y_dense=y)

    def all_reduce_resid_slabs(self, slab: int, S: int, svh: torch.Tensor, resid: torch.Tensor, ss_part: torch.Tensor, m: int):
        """all_reduce_resid for a row-sharded linear launched with a deferred epilogue (ext.exl3_gemv_ex* slabs): the IPC launch finishes the slabs,
        reduces over the ranks and adds to the residual.  Only with the IPC path enabled (callers check `self.ipc`)."""
        assert self.ipc is not None, "all_reduce_resid_slabs needs the IPC all-reduce"
        self.ipc.reduce_slabs(slab, S, svh, resid, ss_part, m)

    # ---- the same boundary for the fx pipeline: the residual stream is a 64-bit fixed-point accumulator R (llama_path.decode_step_fx) ----------
    def all_reduce_fx_slabs(self, slab: int, S: int, svh: torch.Tensor, R: torch.Tensor, m: int):
        """R += sum over ranks of the row-sharded linear whose deferred slabs (slab, S, svh) this rank holds: ONE launch (slab finish + push + rank-order
        sum + add into the fixed-point residual; bit-identical accumulators on every rank).  Only with the IPC path enabled (callers check `self.ipc`)."""
        assert self.ipc is not None, "all_reduce_fx_slabs needs the IPC all-reduce"
        self.ipc.reduce_fx(R, slab=slab, S=S, svh=svh, m=m)

    def all_reduce_fx(self, y: torch.Tensor, R: torch.Tensor, m: int):
        """R += sum over ranks of y (fp32 [m, hidden], this rank's partial rows; overwritten by the library route)."""
        from . import ext
        if self.ipc is not None and y.numel() <= self.ipc.max_elems:
            self.ipc.reduce_fx(R, y=y, m=m)
            return
        self.all_reduce(y)
        ext.fx_add(R, y=y)

    def fwd_barrier(self):
        if self.world_size > 1:
            dist.barrier()

    def all_reduce(self, tensor: torch.Tensor, contribution: bool = True):
        if self.world_size == 1:
            return
        if not contribution:
            tensor.zero_()
        if self.bf16_wire and tensor.dtype == torch.float32:
            t = tensor.to(torch.bfloat16)
            dist.all_reduce(t)
            tensor.copy_(t)
        else:
            dist.all_reduce(tensor)

    def all_reduce_max(self, tensor: torch.Tensor):
        if self.world_size > 1:
            dist.all_reduce(tensor, op=dist.ReduceOp.MAX)

    def broadcast(self, tensor: torch.Tensor, src_rank: int):
        if self.world_size > 1:
            dist.broadcast(tensor, src=src_rank)

    def gather(self, tensor: torch.Tensor, out_tensor: torch.Tensor | None, dst_rank: int, ldims: list[int]):
        """Concatenate per-rank column shards (widths ldims) into out_tensor on dst_rank (modules/gather.py:65)."""
        if self.world_size == 1:
            if out_tensor is not None:
                out_tensor.copy_(tensor)
            return
        if len(set(ldims)) == 1:
            parts = [torch.empty_like(tensor) for _ in range(self.world_size)] if self.rank == dst_rank else None
            dist.gather(tensor, parts, dst=dst_rank)
        else:
            # uneven shards: pad to the widest
            w = max(ldims)
            pad = torch.zeros(tensor.shape[:-1] + (w,), dtype=tensor.dtype, device=tensor.device)
            pad[..., : tensor.shape[-1]] = tensor
            parts = [torch.empty_like(pad) for _ in range(self.world_size)] if self.rank == dst_rank else None
            dist.gather(pad, parts, dst=dst_rank)
        if self.rank == dst_rank:
            od = 0
            for p, ld in zip(parts, ldims):
                out_tensor[..., od: od + ld] = p[..., :ld]
                od += ld



class OneRankOfMany:
    """Stand-in backend for measuring ONE rank's compute leg of a tensor-parallel job on a single GPU (bench.py's `llama-3.1-70b_tp8_rank` line,
    tests/test_gpu_path.py::test_tp_rank_code_path_on_one_gpu): rank 0 of a world of `world_size` whose collectives are no-ops -- the model builds
    that rank's column / row shards and runs the TP branch of the decode pipeline on its own partial sums.  NOT a collective backend: the numbers
    it produces are per-rank partial results; the all-reduce is priced separately (bench.py adds a stated estimate)."""

    def __init__(self, world_size: int, device: torch.device | None = None):
        self.rank, self.world_size, self.device, self.ipc, self.calls = 0, world_size, device, None, 0

    def all_reduce(self, tensor, contribution=True): self.calls += 1
    def all_reduce_max(self, tensor): pass
    def fwd_barrier(self): pass
    def close(self): pass

    def all_reduce_resid(self, y, resid, ss_part, m):
        # TPBackendRCCL.all_reduce_resid's library route (all_reduce + glue_resid) with the exchange left out
        from . import ext
        self.calls += 1
        ext.glue_resid(None, 0, None, None, resid, ss_part, m, y_dense=y)

    ipc_like = True                     # (llama_path: take the slab route, as with the IPC all-reduce enabled)

    def all_reduce_fx_slabs(self, slab, S, svh, R, m):
        # the IPC route's one launch per boundary (slab finish + exchange + add into the fixed-point residual) with the exchange left out
        from . import ext
        self.calls += 1
        ext.fx_add(R, slab=slab, S=S, svh=svh)

    def all_reduce_fx(self, y, R, m):
        from . import ext
        self.calls += 1
        ext.fx_add(R, y=y)


class IpcAllReduce:
    """Host side of exl3_allreduce.hip: this rank's receive buffer + the peers' buffers mapped through hipIpc handles exchanged over the process
    group.  Every rank must issue the same sequence of reduce() calls (a captured hipGraph replays them identically)."""

    MAX_ELEMS = 8 * 512 * 128          # exl3_allreduce.hip: 8 * EXL3_AR_MAX_WGS tasks x 128 values (the grid has to be co-resident)

    def __init__(self, rank: int, world: int, device: torch.device, max_elems: int):
        from . import _lib
        self._lib = _lib.lib()
        # exl3_ar_create clamps to the one-shot push's launch bound (8 * 512 tasks of 128 values): larger messages take the collective library
        self.rank, self.world, self.device, self.max_elems = rank, world, device, min(int(max_elems), self.MAX_ELEMS)
        self.ctx = None

    @classmethod
    def create(cls, rank: int, world: int, device: torch.device, max_elems: int, agree):
        """Collective constructor: allocate, exchange the IPC handles, map the peers.  `agree(ok) -> bool` is an all-ranks AND (a collective);
        the sequence of collectives is the same on every rank whatever fails locally.  Returns None on every rank if any rank failed."""
        from . import _lib
        self = cls(rank, world, device, max_elems)
        handle = ctypes.create_string_buffer(64)
        ok = True
        try:
            torch.cuda.set_device(device)
            ctx = ctypes.c_void_p()
            _lib.check(self._lib.exl3_ar_create(world, rank, self.max_elems, ctypes.byref(ctx), handle))
            self.ctx = ctx
        except Exception:
            ok = False
        handles = [None] * world
        dist.all_gather_object(handles, (ok, bytes(handle.raw)))
        ok = ok and all(h[0] for h in handles)
        if ok:
            try:
                for r in range(world):
                    if r != rank:
                        _lib.check(self._lib.exl3_ar_open_peer(self.ctx, r, ctypes.create_string_buffer(handles[r][1], 64)))
            except Exception:
                ok = False
        if not agree(ok):                                                   # also the barrier: nobody pushes before every mapping exists
            self.close()
            return None
        return self

    def reduce(self, y: torch.Tensor, y_out: torch.Tensor | None = None, resid: torch.Tensor | None = None, ss_part: torch.Tensor | None = None,
               m: int | None = None):
        from . import _lib
        assert y.dtype == torch.float and y.is_contiguous() and y.is_cuda
        hidden = y.shape[-1]
        rows = m if m is not None else y.numel() // hidden
        p = lambda t: None if t is None else t.data_ptr()
        _lib.check(self._lib.exl3_ar_reduce(self.ctx, p(y), p(y_out), p(resid), p(ss_part), rows, hidden, torch.cuda.current_stream(y.device).cuda_stream))

    def reduce_slabs(self, slab: int, S: int, svh: torch.Tensor, resid: torch.Tensor, ss_part: torch.Tensor, m: int, y_out: torch.Tensor | None = None):
        from . import _lib
        hidden = resid.shape[-1] if resid is not None else y_out.shape[-1]
        p = lambda t: None if t is None else t.data_ptr()
        dev = resid.device if resid is not None else y_out.device
        _lib.check(self._lib.exl3_ar_reduce_slabs(self.ctx, None, slab, int(S), p(svh), p(y_out), p(resid), p(ss_part), int(m), hidden,
                                                  torch.cuda.current_stream(dev).cuda_stream))

    def reduce_fx(self, R: torch.Tensor, y: torch.Tensor | None = None, slab: int = 0, S: int = 0, svh: torch.Tensor | None = None, m: int | None = None):
        """R (int64 fixed-point accumulator [m, hidden]) += sum over ranks of y (fp32 rows) or of the finished deferred slabs (slab, S, svh)."""
        from . import _lib
        assert R.dtype == torch.long and R.is_contiguous() and ((y is not None) != bool(slab))
        hidden = R.shape[-1]
        rows = m if m is not None else R.numel() // hidden
        p = lambda t: None if t is None else t.data_ptr()
        _lib.check(self._lib.exl3_ar_reduce_fx(self.ctx, p(y), ctypes.c_void_p(slab) if slab else None, int(S), p(svh), p(R), rows, hidden,
                                               torch.cuda.current_stream(R.device).cuda_stream))

    def epoch(self) -> int:
        from . import _lib
        v = ctypes.c_uint32(0)
        _lib.check(self._lib.exl3_ar_epoch(self.ctx, ctypes.byref(v), torch.cuda.current_stream(self.device).cuda_stream))
        return int(v.value)

    def error(self) -> int:
        from . import _lib
        return _lib.check(self._lib.exl3_ar_error(self.ctx, torch.cuda.current_stream(self.device).cuda_stream))

    def close(self):
        if self.ctx is not None:
            try:
                torch.cuda.synchronize(self.device)
            except Exception:
                pass
            self._lib.exl3_ar_destroy(self.ctx)
            self.ctx = None
