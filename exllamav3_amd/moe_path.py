"""
Sparse-MoE MLP block of the hot path (BASELINE config 5: Mixtral 8x7B shapes), decode form: router -> indexed / weighted exl3_mgemm.

Mirrors what the reference's BlockSparseMLP does at bsz 1 (modules/block_sparse_mlp.py:51-93 routing_std, then exl3_mgemm with pointer
tables + indices for gate / up and indices + weights for down, libtorch/blocksparse_mlp.cpp), composed from this build's C-ABI ops:

    routing_std                      hidden @ gate -> top-k experts + softmax weights                         (1 launch)
    exl3_mgemm (indices)             gate AND up of the selected experts in ONE launch: the pointer tables hold
                                     [gate_0..gate_E-1, up_0..up_E-1] and the index list is [sel, sel + E]       (1 launch)
    silu_mul                         a = silu(g) * u per selected expert                                      (1 launch)
    exl3_mgemm (indices + weights)   down of the selected experts, routing weight folded into the output scale,
                                     slot-group reduction into the token's row                                (1-3 launches)
"""
from __future__ import annotations
import math
import torch
from . import ext
from .llama_path import _rand_linear


class SyntheticEXL3MoE:
    def __init__(self, hidden: int, inter: int, experts: int, top_k: int, K: int = 4, cb: int = 2, device="cuda:0", seed: int = 0):
        self.hidden, self.inter, self.E, self.top_k, self.K, self.cb = hidden, inter, experts, top_k, K, cb
        self.device = torch.device(device)
        gen = torch.Generator(device=self.device); gen.manual_seed(seed)
        self.gate = [_rand_linear(hidden, inter, K, cb, self.device, gen) for _ in range(experts)]
        self.up = [_rand_linear(hidden, inter, K, cb, self.device, gen) for _ in range(experts)]
        self.down = [_rand_linear(inter, hidden, K, cb, self.device, gen, out_dtype=torch.float) for _ in range(experts)]
        self.router = (torch.randn((hidden, experts), device=self.device, generator=gen) / math.sqrt(hidden)).half()
        ptr = lambda ts: torch.tensor([t.data_ptr() for t in ts], dtype=torch.long, device=self.device)
        gu = self.gate + self.up
        self.gu_B, self.gu_suh, self.gu_svh = ptr([l.trellis for l in gu]), ptr([l.suh for l in gu]), ptr([l.svh for l in gu])
        self.d_B, self.d_suh, self.d_svh = ptr([l.trellis for l in self.down]), ptr([l.suh for l in self.down]), ptr([l.svh for l in self.down])
        self._state = None

    def alloc_state(self, tokens: int = 1):
        dev, f16 = self.device, torch.half
        t, k = tokens, self.top_k
        self.scores = torch.empty((t, self.E), dtype=f16, device=dev)
        self.sel = torch.empty((t, k), dtype=torch.long, device=dev)
        self.w = torch.empty((t, k), dtype=f16, device=dev)
        self.sel2 = torch.empty((2, t * k), dtype=torch.long, device=dev)      # [gate slots | up slots]
        self.gu = torch.empty((2 * t * k, 1, self.inter), dtype=f16, device=dev)
        self.a = torch.empty((t * k, 1, self.inter), dtype=f16, device=dev)
        self.d = torch.empty((t * k, 1, self.hidden), dtype=torch.float, device=dev)
        self._state = t

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: (tokens, hidden) fp16 (already normalised).  Returns (tokens, hidden) fp32 = sum_k w_k * expert_k(x)."""
        t, k = x.shape[0], self.top_k
        if self._state != t:
            self.alloc_state(t)
        ext.routing_std(x, self.router, self.scores, self.sel, self.w)
        mcg, mul1 = self.cb == 1, self.cb == 2
        if t == 1:
            # one shared input row: gate and up of the k selected experts in one launch (slots = 2k)
            torch.stack((self.sel.view(-1), self.sel.view(-1) + self.E), out=self.sel2)
            ext.exl3_mgemm(x.view(1, 1, -1), self.gu_B, self.gu, self.gu_suh, None, self.gu_svh, self.sel2.view(-1), None,
                           self.K, -1, mcg, mul1, -1, -1, 0)
        else:
            # several tokens: every (token, expert) slot has its own input row
            xs = x.repeat_interleave(k, dim=0).view(t * k, 1, -1).contiguous()
            sel = self.sel.view(-1)
            ext.exl3_mgemm(xs, self.gu_B, self.gu[: t * k], self.gu_suh, None, self.gu_svh, sel, None, self.K, -1, mcg, mul1, -1, -1, 0)
            ext.exl3_mgemm(xs, self.gu_B, self.gu[t * k:], self.gu_suh, None, self.gu_svh, (sel + self.E).contiguous(), None,
                           self.K, -1, mcg, mul1, -1, -1, 0)
        ext.silu_mul(self.gu[: t * k], self.gu[t * k:], self.a)
        ext.exl3_mgemm(self.a, self.d_B, self.d, self.d_suh, None, self.d_svh, self.sel.view(-1), self.w.view(-1), self.K, -1, mcg, mul1,
                       -1, -1, 0, num_tokens=t)
        return self.d[:t, 0]

    def packed_bytes_per_token(self) -> int:
        """Algorithmic weight bytes one token touches: top_k experts x (gate + up + down)."""
        return self.top_k * 3 * self.hidden * self.inter * self.K // 8
