"""
Sparse-MoE MLP block of the hot path (BASELINE config 5: Mixtral 8x7B shapes): decode form (router -> indexed / weighted exl3_mgemm) and, round 4,
the prefill form (forward_prefill: assignments grouped by expert, per-expert reconstruct + GEMM, fixed-order weighted scatter).

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
import os
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
        self.first, self.last = 0, experts
        self._build_tables()

    #: forward_prefill: the block as two grouped GEMMs over all experts (ext.gemm_nt_grouped) where every expert is local and the shapes fit the own kernel; False:
    #: the per-expert loop (the reference's structure: one GEMM pair per expert, sized on the host)
    grouped_prefill = True

    # k-slices of the fx block's gate|up / down launches (0: the dispatcher / the rule in forward_fx); EXL3_HIP_MOE_FX_SPLIT="g,d" overrides (tuning)
    fx_split = tuple(int(v) for v in os.environ.get("EXL3_HIP_MOE_FX_SPLIT", "0,0").split(","))

    def _build_tables(self):
        """Device pointer tables of the LOCAL experts [first, last) (all of them without expert parallelism)."""
        ptr = lambda ts: torch.tensor([t.data_ptr() for t in ts], dtype=torch.long, device=self.device)
        gu = self.gate + self.up
        self.gu_B, self.gu_suh, self.gu_svh = ptr([l.trellis for l in gu]), ptr([l.suh for l in gu]), ptr([l.svh for l in gu])
        self.g_B, self.g_suh, self.g_svh = ptr([l.trellis for l in self.gate]), ptr([l.suh for l in self.gate]), ptr([l.svh for l in self.gate])
        self.u_B, self.u_suh, self.u_svh = ptr([l.trellis for l in self.up]), ptr([l.suh for l in self.up]), ptr([l.svh for l in self.up])
        self.d_B, self.d_suh, self.d_svh = ptr([l.trellis for l in self.down]), ptr([l.suh for l in self.down]), ptr([l.svh for l in self.down])
        self._state = None

    # ---- checkpoints + expert parallelism (each rank holds experts [first, last), the kernel filters the routed indices to
    #      that range -- modules/block_sparse_mlp.py:1556-1590, quant/exl3_gemm_kernel.cuh:99-127 -- and the partial sums are all-reduced)
    def save_checkpoint(self, directory: str, prefix: str = "model.layers.0.block_sparse_moe") -> None:
        """Mixtral tensor names: prefix.gate.weight (router, (experts, hidden)), prefix.experts.{e}.w1 / w3 / w2 = gate / up / down."""
        import os
        from safetensors.torch import save_file
        assert (self.first, self.last) == (0, self.E), "save the whole block"
        os.makedirs(directory, exist_ok=True)
        t = {prefix + ".gate.weight": self.router.t().contiguous().cpu()}
        for e in range(self.E):
            for nm, lin in (("w1", self.gate[e]), ("w3", self.up[e]), ("w2", self.down[e])):
                key = f"{prefix}.experts.{e}.{nm}"
                t[key + ".trellis"] = lin.trellis.cpu(); t[key + ".suh"] = lin.suh.cpu(); t[key + ".svh"] = lin.svh.cpu()
                if lin.mul1: t[key + ".mul1"] = torch.zeros(1, dtype=torch.int32)
                if lin.mcg: t[key + ".mcg"] = torch.zeros(1, dtype=torch.int32)
        save_file(t, os.path.join(directory, "moe.safetensors"))

    @classmethod
    def from_checkpoint(cls, directory: str, top_k: int, prefix: str = "model.layers.0.block_sparse_moe", device="cuda:0",
                        first_expert: int = 0, last_expert: int | None = None) -> "SyntheticEXL3MoE":
        """One sparse-MoE block from an EXL3 checkpoint directory.  first_expert / last_expert: this rank's expert range (expert
        parallelism): only those experts' tensors are read; forward() then returns the PARTIAL sum over the local experts among the routed
        ones (all-reduce it across ranks; the router is replicated)."""
        from . import loader
        stc = loader.SafetensorsCollection(directory)
        self = cls.__new__(cls)
        self.device = torch.device(device)
        self.router = stc.get_tensor(prefix + ".gate.weight", self.device).half().t().contiguous()          # (hidden, experts)
        self.hidden, self.E = self.router.shape
        self.top_k = top_k
        self.first, self.last = first_expert, self.E if last_expert is None else last_expert
        assert 0 <= self.first < self.last <= self.E
        load = lambda e, nm, dt=None: loader.load_linear_exl3(stc, f"{prefix}.experts.{e}.{nm}", self.device, dt)
        self.gate = [load(e, "w1") for e in range(self.first, self.last)]
        self.up = [load(e, "w3") for e in range(self.first, self.last)]
        self.down = [load(e, "w2", torch.float) for e in range(self.first, self.last)]
        kinds = {(l.K, l.mcg, l.mul1) for l in self.gate + self.up + self.down}
        if len(kinds) != 1:
            raise NotImplementedError(f"{prefix}: experts mix bits-per-weight / codebooks {sorted(kinds)}; one indexed launch needs one kind")
        l0 = self.gate[0]
        self.inter, self.K, self.cb = l0.out_features, l0.K, (2 if l0.mul1 else (1 if l0.mcg else 0))
        self._build_tables()
        return self

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

    def forward(self, x: torch.Tensor, resid_norm=None, add_into=None):
        """x: (tokens, hidden) fp16 (already normalised).  Returns (tokens, hidden) fp32 = sum_k w_k * expert_k(x).
        resid_norm = (resid, norm_w, ss_part, eps): x is an OUTPUT buffer instead -- the router launch forms rms_norm(resid) itself (mean square
        from the residual kernel's per-block sums), routes on it and leaves it in x for the expert launches (no rms_norm launch).
        add_into = (resid, ss_part) (one rank, all experts local): the weighted down launch is deferred and ONE glue launch finishes it, sums the
        slots and adds the result to the residual stream (ext.glue_resid_moe) instead of split-k reduce + slot sum + residual add; returns None."""
        t, k = x.shape[0], self.top_k
        if self._state != t:
            self.alloc_state(t)
        # the router also leaves the [gate slots | up slots] index list of the one-launch gate|up mgemm (no torch index kernels in between)
        if resid_norm is not None:
            resid, norm_w, ss_part, eps = resid_norm
            ext.routing_std_norm(resid, norm_w, ss_part, eps, x, self.router, self.scores, self.sel, self.w, gu_slots=self.sel2)
        else:
            ext.routing_std(x, self.router, self.scores, self.sel, self.w, gu_slots=self.sel2)
        mcg, mul1 = self.cb == 1, self.cb == 2
        if (self.first, self.last) != (0, self.E):
            return self._forward_expert_parallel(x)
        if t == 1:
            # one shared input row: gate and up of the k selected experts in one launch (slots = 2k)
            ext.exl3_mgemm(x.view(1, 1, -1), self.gu_B, self.gu, self.gu_suh, None, self.gu_svh, self.sel2.view(-1), None,
                           self.K, -1, mcg, mul1, -1, -1, 0)
        else:
            # several tokens: every (token, expert) slot has its own input row
            xs = x.repeat_interleave(k, dim=0).view(t * k, 1, -1).contiguous()
            ext.exl3_mgemm(xs, self.gu_B, self.gu[: t * k], self.gu_suh, None, self.gu_svh, self.sel2[0], None, self.K, -1, mcg, mul1, -1, -1, 0)
            ext.exl3_mgemm(xs, self.gu_B, self.gu[t * k:], self.gu_suh, None, self.gu_svh, self.sel2[1], None, self.K, -1, mcg, mul1, -1, -1, 0)
        # down of the routed experts on a = silu(g) * u, formed while the launch builds its activation fragments (no silu_mul launch)
        if add_into is not None:
            resid, ss_part = add_into
            slab, S = ext.exl3_mgemm_act_deferred(self.gu[: t * k], self.gu[t * k:], self.d_B, self.d_suh, self.sel.view(-1), self.K, mcg, mul1, self.hidden)
            ext.glue_resid_moe(slab, S, self.d_svh, self.sel.view(-1), self.w.view(-1), resid, ss_part, t)
            return None
        ext.exl3_mgemm_act(self.gu[: t * k], self.gu[t * k:], self.d_B, self.d, self.d_suh, self.d_svh, self.sel.view(-1), self.w.view(-1), self.K,
                           mcg, mul1, -1, -1, num_tokens=t)
        return self.d[:t, 0]

    def forward_fx(self, R: torch.Tensor, norm_w: torch.Tensor, eps: float, ss_out: torch.Tensor, xn: torch.Tensor) -> None:
        """The block in the fx decode pipeline, 3 launches, one token, all experts local: R (int64 [1, hidden], the fixed-point residual accumulator)
        += sum_k w_k expert_k(rms_norm(R)).  Router on the RMSNorm it forms from R itself (exact sums of squares -> ss_out for the next layer's
        q|k|v launch), indexed gate|up launch leaving slabs, indexed down launch that finishes silu(g) * u from them and adds its weighted rows into R
        with integer atomics (order-independent: bit-reproducible)."""
        assert R.shape[0] == 1 and (self.first, self.last) == (0, self.E)
        if self._state != 1:
            self.alloc_state(1)
        ext.routing_std_fx(R, norm_w, ss_out, eps, xn, self.router, self.scores, self.sel, self.w, gu_slots=self.sel2)
        mcg, mul1 = self.cb == 1, self.cb == 2
        slab, S = ext.exl3_mgemm_deferred(xn.view(1, 1, -1), self.gu_B, self.gu_suh, self.sel2.view(-1), self.K, mcg, mul1, self.inter, self.fx_split[0])
        sd = self.fx_split[1]
        if sd == 0:
            # down launch: slices of 7 Hadamard blocks, so that the 8 half-waves of a 4-wave workgroup finish the slice's silu(g) * u tasks in ONE round
            # (measured on Mixtral-8x7B, 64 column blocks x 2 slots: S = 8 / 14 / 16 / 19 / 23 -> 368 / 373 / 384 / 375 / 363 tok/s); small blocks keep
            # the dispatcher's chip-balancing choice
            nb = self.inter // 128
            s7 = (nb + 6) // 7
            if s7 * (self.hidden // 128) * self.top_k >= 512:
                sd = s7
        ext.exl3_mgemm_act_fx(slab, S, self.gu_svh, self.E, self.d_B, self.d_suh, self.d_svh, self.sel.view(-1), self.w.view(-1), R, self.K, mcg, mul1,
                              self.inter, m=1, num_tokens=1, force_split=sd)

    def forward_prefill(self, x: torch.Tensor) -> torch.Tensor:
        """The block for a prefill chunk (rows > the decode tiers): the large-batch tier of BlockSparseMLP.forward (modules/block_sparse_mlp.py:1169-1330).
        x (tokens, hidden) fp16, normalised.  Router over all rows -> assignments grouped by expert (stable argsort + bincount; the counts go to
        the host like the reference's expert_count.tolist(): they size the GEMMs) -> per expert with rows: gather its tokens, gate|up as ONE GEMM on
        the stacked reconstructed W^T + silu * mul, down with fp32 output (LinearEXL3's routes: reconstruct_had_slice_t + NT GEMM above 144 rows,
        the small-m kernel below -- the reference's run_single_expert_dq / run_single_expert tiers) -> weighted scatter in ascending assignment order
        (ext.moe_scatter: fixed order, bit-reproducible; the reference scatters with index_add_).  Expert parallelism: only the local experts
        [first, last) run and the result is this rank's partial sum.  Returns (tokens, hidden) fp32."""
        from .linear import LinearEXL3, _same_kind
        T, k, dev = x.shape[0], self.top_k, x.device
        scores = torch.empty((T, self.E), dtype=torch.half, device=dev)
        sel = torch.empty((T, k), dtype=torch.long, device=dev)
        w = torch.empty((T, k), dtype=torch.half, device=dev)
        ext.routing_std(x, self.router, scores, sel, w)
        self.pf_sel, self.pf_w = sel, w                                   # (parity tests read the routing)
        nloc = self.last - self.first
        if (self.grouped_prefill and nloc == self.E and ext.gemm_nt_own_default() and T * k >= 256 and self.hidden % 256 == 0 and self.inter % 128 == 0
                and _same_kind(*self.gate, *self.up) and _same_kind(*self.down) and x.stride(0) % 8 == 0):
            # ---- every expert local: the block as TWO grouped GEMMs (ext.gemm_nt_grouped: one launch walks all experts' tiles; the row boundaries are the cumulative
            # sum of the router's bincount, read by the kernel from device memory -- no expert_count.tolist(), no host synchronisation in the block, capturable).
            # gate|up with silu(g) * u in the epilogue on W^T whose gate / up blocks alternate; down with fp32 output (the reference's fp32 expert outputs in front of
            # its index_add_); same values as the per-expert loop below to the order of the k summation.
            R = T * k
            flat_e = sel.reshape(-1)
            order = torch.argsort(flat_e, stable=True)
            token_sorted = torch.arange(T, device=dev, dtype=torch.long).repeat_interleave(k)[order].contiguous()
            weight_sorted = w.reshape(-1)[order].contiguous()
            rows = torch.zeros((self.E + 1,), dtype=torch.int32, device=dev)
            counts = torch.zeros((self.E,), dtype=torch.int32, device=dev)
            counts.index_add_(0, flat_e, torch.ones_like(flat_e, dtype=torch.int32))            # (torch.bincount sizes its result on the host)
            rows[1:] = torch.cumsum(counts, 0).to(torch.int32)
            xs = x.index_select(0, token_sorted)
            g0 = self.gate[0]
            wt = torch.empty((self.E, 2 * self.inter, self.hidden), dtype=torch.half, device=dev)
            for e in range(self.E):
                ext.reconstruct_had_multi_t(wt[e], [self.gate[e].trellis, self.up[e].trellis], [self.gate[e].suh, self.up[e].suh], [self.gate[e].svh, self.up[e].svh],
                                            g0.K, g0.mcg, g0.mul1, True)
            a = torch.empty((R, self.inter), dtype=torch.half, device=dev)
            ext.gemm_nt_grouped(xs, wt, a, rows, 2)
            del wt
            d0 = self.down[0]
            wd = torch.empty((self.E, self.hidden, self.inter), dtype=torch.half, device=dev)
            for e in range(self.E):
                ext.reconstruct_had_slice_t(wd[e], self.down[e].trellis, self.down[e].suh, self.down[e].svh, d0.K, d0.mcg, d0.mul1, 0)
            D = torch.empty((R, self.hidden), dtype=torch.float, device=dev)
            ext.gemm_nt_grouped(a, wd, D, rows, 3)
            del wd
            out = torch.zeros((T, self.hidden), dtype=torch.float, device=dev)
            ext.moe_scatter(D, torch.arange(R, dtype=torch.int32, device=dev), token_sorted, weight_sorted, out)
            return out
        flat_e = sel.reshape(-1) - self.first
        flat_e = torch.where((flat_e >= 0) & (flat_e < nloc), flat_e, torch.full_like(flat_e, nloc))      # non-local experts: sentinel group, skipped
        flat_t = torch.arange(T, device=dev, dtype=torch.long).repeat_interleave(k)
        order = torch.argsort(flat_e, stable=True)
        token_sorted = flat_t[order].contiguous()
        weight_sorted = w.reshape(-1)[order].contiguous()
        counts = torch.bincount(flat_e, minlength=nloc + 1).tolist()
        n_local = sum(counts[:nloc])
        out = torch.zeros((T, self.hidden), dtype=torch.float, device=dev)
        if n_local == 0:
            return out
        D = torch.empty((n_local, self.hidden), dtype=torch.float, device=dev)
        start = 0
        for e in range(nloc):
            c = counts[e]
            if c == 0:
                continue
            xs = x.index_select(0, token_sorted[start: start + c])
            a = LinearEXL3.forward_gate_up_silu(self.gate[e], self.up[e], xs)
            D[start: start + c] = self.down[e].forward(a, out_dtype=torch.float)
            start += c
        rowmap = torch.arange(T * k, dtype=torch.int32, device=dev)
        if n_local < T * k:
            rowmap[n_local:] = -1                                          # assignments of the other ranks' experts
        ext.moe_scatter(D, rowmap, token_sorted, weight_sorted, out)
        return out

    def _forward_expert_parallel(self, x: torch.Tensor) -> torch.Tensor:
        """Local experts [first, last) only: the launches filter the routed indices to the range (in-range slots are compacted to the front,
        the same order in the gate, up and down launches) and the weighted down launch sums the in-range slots only (a token with no expert on
        this rank gets zeros).  One token per set of launches (the range filter works on one token's slots, as in the reference)."""
        t, k = x.shape[0], self.top_k
        mcg, mul1 = self.cb == 1, self.cb == 2
        for i in range(t):
            xi = x[i].view(1, 1, -1)
            sel, w = self.sel[i].contiguous(), self.w[i].contiguous()
            g, u, a, d = self.gu[i * k: (i + 1) * k], self.gu[(t + i) * k: (t + i + 1) * k], self.a[i * k: (i + 1) * k], self.d[i * k: (i + 1) * k]
            ext.exl3_mgemm(xi, self.g_B, g, self.g_suh, None, self.g_svh, sel, None, self.K, -1, mcg, mul1, self.first, self.last, 0)
            ext.exl3_mgemm(xi, self.u_B, u, self.u_suh, None, self.u_svh, sel, None, self.K, -1, mcg, mul1, self.first, self.last, 0)
            ext.exl3_mgemm_act(g, u, self.d_B, d, self.d_suh, self.d_svh, sel, w, self.K, mcg, mul1, self.first, self.last)
        return self.d.view(t, k, 1, self.hidden)[:, 0, 0]

    def packed_bytes_per_token(self) -> int:
        """Algorithmic weight bytes one token touches: top_k experts x (gate + up + down)."""
        return self.top_k * 3 * self.hidden * self.inter * self.K // 8
