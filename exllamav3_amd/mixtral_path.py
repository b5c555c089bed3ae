"""
The EXL3 hot path of a Mixtral-shaped decoder (BASELINE.json config 5: Mixtral 8x7B, 4.0 bpw, 4-bit KV cache, TP = 2), decode form:
every layer = attention sublayer (q|k|v GEMV with the RMSNorm inside, RoPE + quantized KV append, optional decode attention straight from
the quantized cache, o_proj) + sparse-MoE sublayer (RMSNorm, router, indexed gate|up exl3_mgemm of the routed experts, silu * mul, weighted
down exl3_mgemm) -- reference: TransformerBlock -> Attention (modules/attn.py) + BlockSparseMLP.forward (modules/block_sparse_mlp.py:1099-1478).

Parallelism over N ranks as the reference does it for this model: attention is tensor-parallel (column shards of q/k/v by whole
KV-head groups, row shard of o_proj + one all-reduce), the MoE block is expert-parallel (each rank holds experts [first, last), the routed
indices outside the range are filtered inside the launches, the partial sums are all-reduced).  Both all-reduces go through
TPBackendRCCL.all_reduce_resid: the one-shot IPC push fused with the residual add where it is enabled, RCCL + glue_resid otherwise.
"""
from __future__ import annotations
from dataclasses import dataclass
import torch
from . import ext
from .llama_path import _rand_linear
from .moe_path import SyntheticEXL3MoE
from .tp import TPBackendRCCL


@dataclass(frozen=True)
class MixtralShape:
    name: str
    hidden: int
    inter: int
    layers: int
    heads_q: int
    heads_kv: int
    head_dim: int
    vocab: int
    experts: int
    top_k: int
    rope_theta: float = 1000000.0

    def linear_shapes(self):
        """(k, n) of the quantized linears of one layer (one expert's gate / up / down) + lm_head, keyed like LlamaShape.linear_shapes."""
        h, i, hd = self.hidden, self.inter, self.head_dim
        return {"q": (h, self.heads_q * hd), "k": (h, self.heads_kv * hd), "v": (h, self.heads_kv * hd), "o": (self.heads_q * hd, h),
                "gate": (h, i), "up": (h, i), "down": (i, h), "lm_head": (h, self.vocab)}

    def decode_bytes_per_token(self, K: int) -> int:
        """Algorithmic bytes of the quantized linears one token touches at bs = 1: attention linears + top_k experts x (gate, up, down) per layer
        (k * n * K / 8 + 2 (k + n) each) + the fp16 router + lm_head."""
        h, i, hd = self.hidden, self.inter, self.head_dim
        lin = lambda k, n: k * n * K // 8 + 2 * (k + n)
        attn = lin(h, self.heads_q * hd) + 2 * lin(h, self.heads_kv * hd) + lin(self.heads_q * hd, h)
        moe = self.top_k * (2 * lin(h, i) + lin(i, h)) + 2 * h * self.experts
        return (attn + moe) * self.layers + lin(h, self.vocab)


MIXTRAL_8X7B = MixtralShape("mixtral-8x7b", 4096, 14336, 32, 32, 8, 128, 32000, 8, 2)
MIXTRAL_SHAPES = {MIXTRAL_8X7B.name: MIXTRAL_8X7B}


class SyntheticEXL3Mixtral:
    def __init__(self, shape: MixtralShape, K: int = 4, cb: int = 2, device="cuda:0", backend: TPBackendRCCL | None = None, kv_bits: int = 4,
                 seed: int = 0, max_ctx: int = 4096, layers: int | None = None):
        self.shape, self.K, self.cb, self.kv_bits = shape, K, cb, kv_bits
        self.device = torch.device(device)
        self.backend = backend or TPBackendRCCL(0, 1, self.device)
        self.tp, self.rank = self.backend.world_size, self.backend.rank
        self.n_layers = layers or shape.layers
        tp, rank, hd, h = self.tp, self.rank, shape.head_dim, shape.hidden
        assert shape.heads_kv % tp == 0 and shape.experts % tp == 0, "TP degree must divide the KV heads and the experts"
        self.hq, self.hkv = shape.heads_q // tp, shape.heads_kv // tp
        self.first_expert, self.last_expert = rank * shape.experts // tp, (rank + 1) * shape.experts // tp
        gen = torch.Generator(device=self.device); gen.manual_seed(seed * 1000 + rank)
        self.layers = []
        for li in range(self.n_layers):
            moe = SyntheticEXL3MoE(h, shape.inter, shape.experts, shape.top_k, K, cb, self.device, seed=seed * 100 + li)   # same experts / router on every rank
            if tp > 1:
                # expert parallelism: keep this rank's experts only (the others' tensors are freed), launches filter the routed indices to the range
                moe.gate, moe.up, moe.down = (l[self.first_expert:self.last_expert] for l in (moe.gate, moe.up, moe.down))
                moe.first, moe.last = self.first_expert, self.last_expert
                moe._build_tables()
            self.layers.append({
                "q": _rand_linear(h, self.hq * hd, K, cb, self.device, gen), "k": _rand_linear(h, self.hkv * hd, K, cb, self.device, gen),
                "v": _rand_linear(h, self.hkv * hd, K, cb, self.device, gen),
                "o": _rand_linear(self.hq * hd, h, K, cb, self.device, gen, out_dtype=torch.float),
                "norm1": (1.0 + 0.05 * torch.randn(h, device=self.device, generator=gen)).half(),
                "norm2": (1.0 + 0.05 * torch.randn(h, device=self.device, generator=gen)).half(),
                "moe": moe,
            })
            torch.cuda.empty_cache() if tp > 1 else None
        self.final_norm = (1.0 + 0.05 * torch.randn(h, device=self.device, generator=gen)).half()
        vloc = shape.vocab // tp // 128 * 128 if tp > 1 else shape.vocab
        self.vocab_local = vloc
        self.lm_head = _rand_linear(h, vloc, K, cb, self.device, gen)
        self.inv_freq = (1.0 / (shape.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))).to(self.device)
        self.eps, self.page, self.max_ctx = 1e-5, 256, max_ctx
        self.with_attention = False
        self.attn_merge_in_oproj = True                                # fx step, head_dim 128: attention merge inside o_proj's launch (ext.exl3_gemv_ex_attm)
        self.attn_qkv_in_split = True                                  # ... and the q|k|v epilogue inside the context-split launch (ext.attn_decode_qcache_split_qkv)
        self.norm_in_router = True                                     # False: separate rms_norm launch in front of every MoE block
        self.fused_moe_tail = True                                     # one rank: split-k reduce + slot sum + residual add of a MoE block in one launch
        self._state_bsz = None

    def alloc_state(self, bsz: int, pos: int = 1000):
        dev, s, hd = self.device, self.shape, self.shape.head_dim
        G = self.hkv * hd // 32
        pps = self.max_ctx // self.page
        self.block_table = torch.arange(bsz * pps, dtype=torch.int32, device=dev).view(bsz, pps)
        self.positions = torch.full((bsz,), pos, dtype=torch.int32, device=dev)
        n_pages = bsz * pps
        mk = lambda: (torch.zeros((n_pages, self.page, G * self.kv_bits), dtype=torch.int32, device=dev), torch.zeros((n_pages, self.page, G), dtype=torch.half, device=dev))
        self.kcache = [mk() for _ in range(self.n_layers)]
        self.vcache = [mk() for _ in range(self.n_layers)]
        f16, f32 = torch.half, torch.float
        self.x = torch.randn((bsz, s.hidden), device=dev).to(f16)
        self.x0 = self.x.clone()
        self.xn = torch.empty_like(self.x)
        self.q = torch.empty((bsz, 1, self.hq, hd), dtype=f16, device=dev)
        self.o = torch.empty((bsz, s.hidden), dtype=f32, device=dev)
        self.ss = torch.empty((bsz, s.hidden // 128), dtype=f32, device=dev)
        self.ss2 = torch.empty_like(self.ss)
        self.R = torch.zeros((bsz, s.hidden), dtype=torch.long, device=dev)      # fx pipeline: the residual stream in 64-bit fixed point (value * 2^32)
        self.logits = torch.empty((bsz, self.vocab_local), dtype=f16, device=dev)
        self.attn_pos = pos
        self.attn_out = torch.empty((bsz, self.hq, hd), dtype=f16, device=dev)
        self.rope_sin = torch.empty((bsz, 64), dtype=f32, device=dev); self.rope_cos = torch.empty((bsz, 64), dtype=f32, device=dev)
        self.kv_slots = torch.empty((bsz,), dtype=torch.long, device=dev)
        self.attn_lens = torch.full((bsz,), pos + 1, dtype=torch.int32, device=dev)
        self.attn_ws = torch.empty((bsz * self.hq * ((pos + 32) // 32) * 132,), dtype=f32, device=dev)
        for L in self.layers:
            L["moe"].alloc_state(bsz)
        self._state_bsz = bsz

    def decode_step_fx(self):
        """Decode step with the residual stream in a fixed-point accumulator (llama_path.decode_step_fx), batch 1, one rank: 6 launches per layer
        instead of 8 -- q|k|v [reads R, RMSNorm inside], glue_qkv_rs, o_proj [adds into R], router [reads R, exact RMSNorm], indexed gate|up, indexed
        down [silu(g) * u from the slabs, weighted rows added into R]: 3 launches per MoE block, no split-k reduce / slot sum / residual launch."""
        bsz, hd, hidden = self._state_bsz, self.shape.head_dim, self.shape.hidden
        kinds = all(len({(l.K, l.mcg, l.mul1) for l in (L["q"], L["k"], L["v"])}) == 1 for L in self.layers)
        if self.tp != 1 or bsz != 1 or not kinds:
            return self.decode_step()
        R, sc, so_ = self.R, self.ss, self.ss2
        ATOM = ext.GEMV_OUT_ATOMIC
        q2 = self.q.view(bsz, -1)
        # the step's two set-up launches (fixed-point copy of the input row, rope tables + cache rows) as one
        ext.fx_init_prep(self.x0, R, sc, bsz, self.inv_freq, self.positions, hd, self.block_table, self.page, self.rope_sin, self.rope_cos, self.kv_slots)
        tab = (self.rope_sin, self.rope_cos, self.kv_slots)
        for li, L in enumerate(self.layers):
            lq, lk, lv, lo, moe = L["q"], L["k"], L["v"], L["o"], L["moe"]
            kc, ks = self.kcache[li]
            vc, vs = self.vcache[li]
            slabs, S = ext.exl3_gemv_ex_fx(R, L["norm1"], sc, so_, self.eps, [lq.trellis, lk.trellis, lv.trellis], [lq.suh, lk.suh, lv.suh],
                                           bsz, lq.mcg, lq.mul1)
            so_split = 8 if self.hq * hd == 4096 else 0
            fuse_qkv = self.with_attention and hd == 128 and self.attn_merge_in_oproj and self.attn_qkv_in_split and tab is not None and self.kv_bits == 4
            if fuse_qkv:
                # q|k|v epilogue inside the attention's context-split launch, merge inside o_proj's (llama_path.decode_step_fx): 3 launches for the sublayer
                ns, _ = ext.attn_decode_qcache_split_qkv(slabs, S, lq.svh, lk.svh, lv.svh, self.q.view(bsz, self.hq, hd), self.inv_freq, self.positions,
                                                         kc, ks, vc, vs, self.block_table, self.attn_lens, self.attn_pos + 1, self.attn_ws, tab,
                                                         sc, so_, hidden, self.eps)
            else:
                ext.glue_qkv_rs(slabs, S, lq.svh, lk.svh, lv.svh, self.q, None, None, self.inv_freq, self.positions, kc, ks, vc, vs,
                                self.block_table, self.page, self.kv_bits, self.kv_bits, bsz, self.hq, self.hkv, hd, sc, so_, hidden, self.eps, tab=tab)
            sc, so_ = so_, sc
            o_in = q2
            if fuse_qkv:
                ext.exl3_gemv_ex_attm(self.attn_ws, ns, self.hq, self.hkv, lo.trellis, R, lo.suh, lo.svh, bsz, lo.mcg, lo.mul1, ATOM, so_split)
            elif self.with_attention and hd == 128 and self.attn_merge_in_oproj:
                # the flash-decoding merge of the context splits runs inside o_proj's launch (llama_path.decode_step_fx)
                ns = ext.attn_decode_qcache_split(self.q.view(bsz, self.hq, hd), kc, ks, vc, vs, self.block_table, self.attn_lens, self.attn_pos + 1, self.attn_ws)
                ext.exl3_gemv_ex_attm(self.attn_ws, ns, self.hq, self.hkv, lo.trellis, R, lo.suh, lo.svh, bsz, lo.mcg, lo.mul1, ATOM, so_split)
            else:
                if self.with_attention:
                    ext.attn_decode_qcache(self.q.view(bsz, self.hq, hd), self.attn_out, kc, ks, vc, vs, self.block_table, self.attn_lens,
                                           self.attn_pos + 1, workspace=self.attn_ws)
                    o_in = self.attn_out.view(bsz, -1)
                ext.exl3_gemv_ex(o_in, None, None, [lo.trellis], [R], [lo.suh], [lo.svh], bsz, lo.mcg, lo.mul1, ATOM, so_split)
            moe.forward_fx(R, L["norm2"], self.eps, so_, self.xn)      # the router leaves the exact sums of squares of the residual it read
            sc, so_ = so_, sc
        ext.fx_finish(R, self.x, sc, bsz)
        ext.exl3_gemv_ex_norm(self.x, self.final_norm, sc, self.eps, [self.lm_head.trellis], [self.logits], [self.lm_head.suh], [self.lm_head.svh],
                              bsz, self.lm_head.mcg, self.lm_head.mul1, 0)
        return self.logits

    # ---- one prefill chunk (config 5's prefill leg) ----------------------------------------------------------------
    prefill_attention = False

    def prefill_flops_per_token(self) -> int:
        """2 * (attention linears + top_k x (gate + up + down)) per layer (router and lm_head excluded, as for the Llama shapes)."""
        s, hd = self.shape, self.shape.head_dim
        attn = s.hidden * (s.heads_q * hd) * 2 + s.hidden * (s.heads_kv * hd) * 2
        return 2 * (attn + s.top_k * 3 * s.hidden * s.inter) * self.n_layers

    def prefill_allreduce_dtype(self):
        """dtype of the (tokens, hidden) partial sums a rank hands to the all-reduce in prefill_chunk (o_proj and the MoE block produce fp32)."""
        return torch.float

    def prefill_chunk(self, tokens: int):
        """All linears + norms + rope + KV-quant + the sparse-MoE block of a `tokens`-token chunk (bsz 1): llama_path.prefill_chunk with the MLP replaced
        by SyntheticEXL3MoE.forward_prefill (modules/block_sparse_mlp.py:1169-1330: the grouped-by-expert large-batch tier).  Attention core out of the
        default scope like there (attention output := rope(q)); prefill_attention adds dequant_cache_paged + the causal chunk attention.  Tensor /
        expert parallel ranks all-reduce the o_proj and MoE partial sums."""
        from .linear import LinearEXL3
        s, hd, dev, be = self.shape, self.shape.head_dim, self.device, self.backend
        if getattr(self, "_pf_tokens", None) != tokens:
            self.px0 = torch.randn((tokens, s.hidden), device=dev).half()
            self._pf_tokens = tokens
            G = self.hkv * hd // 32
            pages = (tokens + self.page - 1) // self.page
            self.pf_bt = torch.arange(pages, dtype=torch.int32, device=dev).view(1, pages)
            self.pf_sl = torch.zeros((1,), dtype=torch.int32, device=dev)
            self.pf_cache = [torch.zeros((pages, self.page, G * self.kv_bits), dtype=torch.int32, device=dev) for _ in range(2)] + \
                            [torch.zeros((pages, self.page, G), dtype=torch.half, device=dev) for _ in range(2)]
            self.pf_kd = torch.zeros((pages, self.page, self.hkv * hd), dtype=torch.half, device=dev)
            self.pf_vd = torch.zeros_like(self.pf_kd)
            self.pf_len = torch.full((1,), tokens, dtype=torch.int32, device=dev)
            self.pf_ao = torch.empty((tokens, self.hq * hd), dtype=torch.half, device=dev)
        x = self.px0.clone()
        xn = torch.empty_like(x)
        for L in self.layers:
            ext.rms_norm(x, L["norm1"], xn, self.eps)
            qkv = LinearEXL3.forward_multi([L["q"], L["k"], L["v"]], xn) if hd == 128 else None
            if qkv is not None:
                q, k, v = qkv                                              # column ranges of ONE GEMM's output: rope and the cache append take the row stride
                ext.rope_strided(q, k, self.inv_freq, 0, None, None, 1.0, 1, tokens)
                ext.quant_cache_paged_strided(k, self.pf_cache[0], self.pf_cache[2], v, self.pf_cache[1], self.pf_cache[3], self.pf_sl, self.pf_bt, self.page, tokens)
            else:
                q4_ = L["q"].forward(xn).view(1, tokens, self.hq, hd); k4_ = L["k"].forward(xn).view(1, tokens, self.hkv, hd); v4_ = L["v"].forward(xn).view(1, tokens, self.hkv, hd)
                ext.rope(q4_, q4_, k4_, k4_, self.inv_freq, 0, None, None, 2, 1.0)
                ext.quant_cache_paged(k4_.view(1, tokens, -1), self.pf_cache[0], self.pf_cache[2], v4_.view(1, tokens, -1), self.pf_cache[1], self.pf_cache[3],
                                      self.pf_sl, self.pf_bt, self.page, tokens)
                q = q4_.view(tokens, -1)
            if self.prefill_attention:
                ext.dequant_cache_paged(self.pf_cache[0], self.pf_cache[2], self.pf_kd, self.pf_cache[1], self.pf_cache[3], self.pf_vd, self.pf_len, self.pf_bt, self.page)
                q4 = q.as_strided((1, tokens, self.hq, hd), (tokens * q.stride(0), q.stride(0), hd, 1), q.storage_offset())
                ext.attn_prefill_paged(q4, self.pf_ao.view(1, tokens, self.hq, hd), self.pf_kd.view(-1, self.page, self.hkv, hd),
                                       self.pf_vd.view(-1, self.page, self.hkv, hd), self.pf_bt, self.pf_len)
                q = self.pf_ao
            if self.tp == 1:
                L["o"].forward_add_residual(q, x)
                ext.rms_norm(x, L["norm2"], xn, self.eps)
            else:
                o = L["o"].forward(q.contiguous().view(tokens, -1))
                be.all_reduce(o)
                ext.rms_norm_res_in(o, L["norm2"], xn, x, self.eps)
            y = L["moe"].forward_prefill(xn)
            if self.tp > 1:
                be.all_reduce(y)
            ext.add(x, y)
        self.px_out = x
        ext.rms_norm(x[-1:], self.final_norm, xn[-1:], self.eps)
        return self.lm_head.forward(xn[-1:].contiguous())

    def decode_step(self):
        """One decode step (bsz tokens, one per sequence), graph-capturable.  Per layer: 4 launches for the attention sublayer (+2 with the
        attention core), router (RMSNorm inside) + gate|up + down + one tail launch for the MoE block on one rank (3-5 launches + the fused IPC all-reduce with EP)."""
        bsz, hd, be = self._state_bsz, self.shape.head_dim, self.backend
        x, ss = self.x, self.ss
        x.copy_(self.x0)
        DEF = ext.GEMV_OUT_DEFERRED
        q2 = self.q.view(bsz, -1)
        ext.qkv_prep(self.inv_freq, self.positions, hd, self.block_table, self.page, self.rope_sin, self.rope_cos, self.kv_slots)
        tab = (self.rope_sin, self.rope_cos, self.kv_slots)
        ext.glue_resid(None, 0, None, None, x, ss, bsz)
        for li, L in enumerate(self.layers):
            lq, lk, lv, lo, moe = L["q"], L["k"], L["v"], L["o"], L["moe"]
            kc, ks = self.kcache[li]
            vc, vs = self.vcache[li]
            slabs, S = ext.exl3_gemv_ex_norm(x, L["norm1"], ss, self.eps, [lq.trellis, lk.trellis, lv.trellis], None, [lq.suh, lk.suh, lv.suh], None,
                                             bsz, lq.mcg, lq.mul1, DEF)
            ext.glue_qkv(slabs, S, lq.svh, lk.svh, lv.svh, self.q, None, None, self.inv_freq, self.positions, kc, ks, vc, vs,
                         self.block_table, self.page, self.kv_bits, self.kv_bits, bsz, self.hq, self.hkv, hd, tab=tab)
            o_in = q2                                                  # attention core out of the default scope, as in llama_path
            if self.with_attention:
                ext.attn_decode_qcache(self.q.view(bsz, self.hq, hd), self.attn_out, kc, ks, vc, vs, self.block_table, self.attn_lens,
                                       self.attn_pos + 1, workspace=self.attn_ws)
                o_in = self.attn_out.view(bsz, -1)
            if self.tp == 1:
                so, So = ext.exl3_gemv_ex(o_in, None, None, [lo.trellis], None, [lo.suh], None, bsz, lo.mcg, lo.mul1, DEF)
                ext.glue_resid(so[0], So, lo.svh, None, x, ss, bsz)
            elif getattr(be, "ipc", None) is not None and bsz * self.shape.hidden <= be.ipc.max_elems:
                # the IPC all-reduce launch is the row shard's epilogue too (deferred slabs in, residual out), as in llama_path
                so, So = ext.exl3_gemv_ex(o_in, None, None, [lo.trellis], None, [lo.suh], None, bsz, lo.mcg, lo.mul1, DEF)
                be.all_reduce_resid_slabs(so[0], So, lo.svh, x, ss, bsz)
            else:
                lo.bc.run(o_in, self.o)
                be.all_reduce_resid(self.o, x, ss, bsz)
            # sparse-MoE sublayer: xn = rms_norm(x) ; y = sum_k w_k expert_k(xn) (this rank's experts) ; x += all_reduce(y)
            add_into = (x, ss) if (self.tp == 1 and self.fused_moe_tail and bsz <= 4) else None
            if self.norm_in_router:
                y = moe.forward(self.xn, resid_norm=(x, L["norm2"], ss, self.eps), add_into=add_into)    # the router launch forms xn = rms_norm(x) from x and ss
            else:
                ext.rms_norm(x, L["norm2"], self.xn, self.eps)
                y = moe.forward(self.xn, add_into=add_into)            # (bsz, hidden) fp32 view of the block's output buffer
            if add_into is not None:
                pass                                                   # glue_resid_moe already added the block's output to x and refreshed ss
            elif self.tp == 1:
                ext.glue_resid(None, 0, None, None, x, ss, bsz, y_dense=y.contiguous() if not y.is_contiguous() else y)
            else:
                be.all_reduce_resid(y.contiguous() if not y.is_contiguous() else y, x, ss, bsz)
        ext.exl3_gemv_ex_norm(x, self.final_norm, ss, self.eps, [self.lm_head.trellis], [self.logits], [self.lm_head.suh], [self.lm_head.svh],
                              bsz, self.lm_head.mcg, self.lm_head.mul1, 0)
        return self.logits
