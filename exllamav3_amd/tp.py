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
        if self.world_size > 1 and dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()

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
