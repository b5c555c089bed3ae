"""
The EXL3 hot path of a Llama-shaped decoder, driven op by op through exllamav3_amd.ext (i.e. through the C-ABI):
every quantized Linear, RMSNorm, RoPE and the KV-cache quantized append of one decode step / one prefill chunk,
on synthetic random EXL3 tensors (random trellis bits are legal inputs, decode is total over all
2^16 states).  This is the workload bench.py times; it is also what tests/test_gpu_path.py checks against the
oracle layer by layer.

Mirrors the reference call stack of SURVEY.md 3.1 (TransformerBlock -> RMSNorm -> Attention(q,k,v,rope,kv-quant,o)
-> RMSNorm(fused residual) -> GatedMLP(gate/up fused, silu*mul, down)) minus the attention core itself (flash-decoding
over the cache), which SURVEY.md 2.1 marks out of scope: the attention output is taken to be q, so that o_proj has a
correctly shaped, data-dependent input.

Tensor parallelism (tp_size > 1): Megatron column/row shards, one all-reduce after o_proj and one after down_proj
(reference: modules/attn.py:547, modules/mlp.py:770), lm_head column shards + gather.
"""
from __future__ import annotations
from dataclasses import dataclass
import math
import os
import torch
from . import ext
from .linear import LinearEXL3
from .tp import TPBackendRCCL, split_points


@dataclass(frozen=True)
class LlamaShape:
    name: str
    hidden: int
    inter: int
    layers: int
    heads_q: int
    heads_kv: int
    head_dim: int
    vocab: int
    rope_theta: float = 500000.0

    def linear_shapes(self):
        """(k, n) of every quantized linear of one layer + lm_head (science/qgemm_benchmark.py:15-34)."""
        h, i = self.hidden, self.inter
        return {"q": (h, self.heads_q * self.head_dim), "k": (h, self.heads_kv * self.head_dim),
                "v": (h, self.heads_kv * self.head_dim), "o": (self.heads_q * self.head_dim, h),
                "gate": (h, i), "up": (h, i), "down": (i, h), "lm_head": (h, self.vocab)}

    def decode_bytes_per_token(self, K: int, head_K: int | None = None) -> int:
        """Algorithmic bytes of the quantized linears per token at bs = 1: sum [k*n*K/8 + 2(k+n)]."""
        s = self.linear_shapes()
        per_layer = sum(k * n * K // 8 + 2 * (k + n) for name, (k, n) in s.items() if name != "lm_head")
        k, n = s["lm_head"]
        return per_layer * self.layers + k * n * (head_K or K) // 8 + 2 * (k + n)

    def prefill_flops_per_token(self) -> int:
        """2 * sum k*n over the layer linears (attention and lm_head excluded)."""
        s = self.linear_shapes()
        return 2 * sum(k * n for name, (k, n) in s.items() if name != "lm_head") * self.layers


LLAMA_3_2_1B = LlamaShape("llama-3.2-1b", 2048, 8192, 16, 32, 8, 64, 128256)
LLAMA_3_1_8B = LlamaShape("llama-3.1-8b", 4096, 14336, 32, 32, 8, 128, 128256)
LLAMA_3_1_70B = LlamaShape("llama-3.1-70b", 8192, 28672, 80, 64, 8, 128, 128256)
SHAPES = {s.name: s for s in (LLAMA_3_2_1B, LLAMA_3_1_8B, LLAMA_3_1_70B)}


def rope_inv_freq(head_dim: int, rope_theta: float, rope_scaling: dict | None = None) -> torch.Tensor:
    """Per-pair rotation frequencies fp32 [head_dim / 2] from a config.json, with the `rope_scaling` transform the reference applies on the
    host (util/rope.py:187-228): none / "default", "linear" (divide by factor) and "llama3" (Llama-3.1+: wavelengths beyond
    original_max_position_embeddings / low_freq_factor are slowed by `factor`, those below .../high_freq_factor kept, the band between
    interpolated).  Other kinds (yarn, longrope, proportional, mrope) are not on this path: NotImplementedError, never a silent default."""
    inv = 1.0 / (rope_theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    kind = None if not rope_scaling else rope_scaling.get("rope_type", rope_scaling.get("type"))
    if kind in (None, "default"):
        return inv
    if kind == "linear":
        return inv / float(rope_scaling.get("factor", 1.0))
    if kind == "llama3":
        factor = float(rope_scaling.get("factor", 8.0))
        lo_f, hi_f = float(rope_scaling.get("low_freq_factor", 1.0)), float(rope_scaling.get("high_freq_factor", 4.0))
        ctx = float(rope_scaling.get("original_max_position_embeddings", 8192))
        wavelen = 2.0 * math.pi / inv
        slowed = torch.where(wavelen > ctx / lo_f, inv / factor, inv)
        t = (ctx / wavelen - lo_f) / (hi_f - lo_f)                      # 0 at the low-frequency edge of the band, 1 at the high one
        blended = (1.0 - t) * slowed / factor + t * slowed
        band = (wavelen >= ctx / hi_f) & (wavelen <= ctx / lo_f)
        return torch.where(band, blended, slowed)
    raise NotImplementedError(f"rope_scaling type {kind!r} is not supported by this path")


def _rand_linear(k: int, n: int, K: int, cb: int, device, gen: torch.Generator, out_dtype=None, out_scale: float = 0.5) -> LinearEXL3:
    trellis = torch.randint(-32768, 32768, (k // 16, n // 16, 16 * K), dtype=torch.int16, device=device, generator=gen)
    sgn_u = torch.where(torch.rand(k, device=device, generator=gen) < 0.5, -1.0, 1.0)
    sgn_v = torch.where(torch.rand(n, device=device, generator=gen) < 0.5, -1.0, 1.0)
    # magnitudes like real checkpoints: keep y at O(out_scale) for unit-RMS x
    suh = (sgn_u * torch.exp(0.2 * torch.randn(k, device=device, generator=gen))).half()
    svh = (sgn_v * (out_scale / math.sqrt(k)) * torch.exp(0.2 * torch.randn(n, device=device, generator=gen))).half()
    return LinearEXL3(k, n, trellis, suh, svh, mcg=(cb == 1), mul1=(cb == 2), out_dtype=out_dtype)


def _rand_linear_host(k: int, n: int, K: int, cb: int, device, rng, out_dtype=None, out_scale: float = 0.5) -> LinearEXL3:
    """_rand_linear with every value drawn on the HOST from a numpy Generator (PCG64: the same stream on every machine and torch build), then copied
    to the device.  The pinned-logits check of bench.py and tests/golden/make_bench_pins.py (oracle, CPU only) build the same tensors from a seed."""
    import numpy as np
    trellis = torch.from_numpy(rng.integers(-32768, 32768, size=(k // 16, n // 16, 16 * K), dtype=np.int16))
    sgn_u = np.where(rng.random(k) < 0.5, -1.0, 1.0)
    sgn_v = np.where(rng.random(n) < 0.5, -1.0, 1.0)
    suh = torch.from_numpy((sgn_u * np.exp(0.2 * rng.standard_normal(k))).astype(np.float16))
    svh = torch.from_numpy((sgn_v * (out_scale / math.sqrt(k)) * np.exp(0.2 * rng.standard_normal(n))).astype(np.float16))
    return LinearEXL3(k, n, trellis.to(device), suh.to(device), svh.to(device), mcg=(cb == 1), mul1=(cb == 2), out_dtype=out_dtype)


def _same_kind(*lins: LinearEXL3) -> bool:
    """One fused launch takes matrices of one bits-per-weight and codebook (the kernels are compiled per K); the reference fuses q|k|v and
    gate|up under the same test (modules/attn.py:439, modules/mlp.py:635) and otherwise runs one GEMV per matrix.  Its bit allocation moves
    q, k, v together (qgroup key.qkv, modules/attn.py:244-280) and gate, up together (key.gu, modules/mlp.py:537-556); o and down are groups of
    their own, so a fractional-bpw checkpoint has fused linears of two adjacent widths (conversion/allocation.py:131-141)."""
    return len({(l.K, l.mcg, l.mul1) for l in lins}) == 1


class SyntheticEXL3Llama:
    def __init__(self, shape: LlamaShape, K: int = 4, cb: int = 2, device: torch.device | str = "cuda:0",
                 backend: TPBackendRCCL | None = None, kv_bits: int = 4, seed: int = 0, head_K: int | None = None,
                 max_ctx: int = 4096, layers: int | None = None, host_seed: int | None = None, layer_K=None):
        """host_seed: draw every tensor (and alloc_state's input rows) on the host from numpy's PCG64 stream of that seed instead of the device
        generator -- machine-independent tensors, used by the pinned-logits check (bench.py, tests/golden/make_bench_pins.py); TP = 1 only.
        layer_K: optional callable (layer index, "qkv" | "o" | "gu" | "d") -> bits per weight of that fused linear (a qgroup of the reference: the allocator of a
        fractional-bpw conversion bumps whole qgroups by one bit, conversion/allocation.py:131-141); default: K everywhere."""
        self.shape, self.K, self.cb, self.kv_bits = shape, K, cb, kv_bits
        if self.fx_act_in_gemv is None:
            self.fx_act_in_gemv = shape.hidden <= 2048
        self.device = torch.device(device)
        self.backend = backend or TPBackendRCCL(0, 1, self.device)
        self.tp, self.rank = self.backend.world_size, self.backend.rank
        self.n_layers = layers or shape.layers
        tp, rank = self.tp, self.rank
        assert shape.heads_kv % tp == 0, "TP degree must divide the KV heads (attention splits whole KV-head groups)"
        self.hq, self.hkv = shape.heads_q // tp, shape.heads_kv // tp
        hd = shape.head_dim
        self._host_rng = None
        if host_seed is not None:
            import numpy as np
            assert tp == 1, "host_seed models are single-rank"
            self._host_rng = rng = np.random.default_rng(host_seed)
            mk_lin = lambda k_, n_, K_, out_dtype=None: _rand_linear_host(k_, n_, K_, cb, self.device, rng, out_dtype=out_dtype)
            mk_norm = lambda h_: torch.from_numpy((1.0 + 0.05 * rng.standard_normal(h_)).astype(np.float16)).to(self.device)
        else:
            gen = torch.Generator(device=self.device)
            gen.manual_seed(seed * 1000 + rank)
            mk_lin = lambda k_, n_, K_, out_dtype=None: _rand_linear(k_, n_, K_, cb, self.device, gen, out_dtype=out_dtype)
            mk_norm = lambda h_: (1.0 + 0.05 * torch.randn(h_, device=self.device, generator=gen)).half()
        ipts = split_points(shape.inter, tp)
        self.inter_local = ipts[rank + 1] - ipts[rank]
        vpts = split_points(shape.vocab, tp)
        self.vocab_local = vpts[rank + 1] - vpts[rank]
        self.vocab_ldims = [vpts[r + 1] - vpts[r] for r in range(tp)]
        h = shape.hidden
        self.layers = []
        for li in range(self.n_layers):
            kq, ko, kg, kd = (layer_K(li, g) for g in ("qkv", "o", "gu", "d")) if layer_K else (K, K, K, K)
            L = {
                "q": mk_lin(h, self.hq * hd, kq),
                "k": mk_lin(h, self.hkv * hd, kq),
                "v": mk_lin(h, self.hkv * hd, kq),
                # o / down: row shards, fp32 partial sums (architecture/llama.py:95,111 out_dtype = float)
                "o": mk_lin(self.hq * hd, h, ko, out_dtype=torch.float),
                "gate": mk_lin(h, self.inter_local, kg),
                "up": mk_lin(h, self.inter_local, kg),
                "down": mk_lin(self.inter_local, h, kd, out_dtype=torch.float),
                "norm1": mk_norm(h),
                "norm2": mk_norm(h),
            }
            if tp > 1:
                # row shards of ONE quantized linear share its output-side scale vector (modules/quant/exl3.py:300-313: an in-split slices suh and the trellis, svh stays
                # whole): o / down of every rank get the SAME svh (a rank-independent stream) -- what makes "sum the ranks' rotated-basis partial rows, then apply the
                # output Hadamard and svh once" (the persistent step's row edges under TP) equal to "finish every rank's row, then sum" (the all-reduce launches)
                gsh = torch.Generator(device=self.device)
                gsh.manual_seed(seed * 1000 + 7777 + li)
                for nm in ("o", "down"):
                    kk = L[nm].trellis.shape[0] * 16 * tp
                    sg = torch.where(torch.rand(h, device=self.device, generator=gsh) < 0.5, -1.0, 1.0)
                    L[nm].svh.copy_((sg * (0.5 / math.sqrt(kk)) * torch.exp(0.2 * torch.randn(h, device=self.device, generator=gsh))).half())
            self.layers.append(L)
        self.final_norm = mk_norm(h)
        self.lm_head = mk_lin(h, self.vocab_local, head_K or K)
        self.inv_freq = (1.0 / (shape.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))).to(self.device)
        self.eps = 1e-5
        # paged quantized KV cache (cache/quant.py:40-42): (pages, 256, kv_dim/32*bits) int32 + scales fp16
        self.page = 256
        self.max_ctx = max_ctx
        self._state_bsz = None

    # ---- pinned-logits check model -------------------------------------------------------------------------
    PIN_VOCAB, PIN_POS, PIN_SEED = 2048, 700, 20260925

    @classmethod
    def pin_model(cls, shape_name: str, K: int, cb: int, device, bsz: int = 1, ctx: int | None = None) -> "SyntheticEXL3Llama":
        """ONE layer of the named shape + a 2048-column lm_head, every tensor and the input rows drawn on the host from a fixed PCG64 seed (the same
        bits on every machine), state allocated at position 700.  tests/golden/make_bench_pins.py runs the ORACLE over exactly this model on the CPU and
        commits its logits (tests/golden/bench_pins.json; tests/test_bench_pins.py re-derives them from the oracle); bench.py runs its timed pipeline over
        it on the GPU and compares -- the check behind the headline number that `isfinite(logits)` used to stand in for (VERDICT r3 weak #1 d)."""
        base = SHAPES[shape_name]
        shape = LlamaShape(base.name + "-pin", base.hidden, base.inter, 1, base.heads_q, base.heads_kv, base.head_dim, cls.PIN_VOCAB, base.rope_theta)
        if ctx is None:
            model = cls(shape, K=K, cb=cb, device=device, kv_bits=4, max_ctx=1024, host_seed=cls.PIN_SEED + 131 * K + cb)
            model.alloc_state(bsz, pos=cls.PIN_POS)
            return model
        # ... WITH the decode attention (round 6: the gate of bench.py's ..._with_attention_ctx* lines): the new token at position ctx behind a pre-filled 4-bit cache whose
        # words and group scales are drawn from the same host stream (uniform levels, scales in [0.05, 0.55)): the cache IS the data, no quantizer involved
        import numpy as np
        model = cls(shape, K=K, cb=cb, device=device, kv_bits=4, max_ctx=(ctx + 1 + 255) // 256 * 256, host_seed=cls.PIN_SEED + 131 * K + cb + 7919 * ctx)
        model.alloc_state(bsz, pos=ctx)
        model.with_attention = True
        rng = model._host_rng
        for c, s_ in model.kcache + model.vcache:
            c.copy_(torch.from_numpy(rng.integers(0, 2 ** 32, size=tuple(c.shape), dtype=np.uint32).view(np.int32)).to(c.device))
            s_.copy_(torch.from_numpy((rng.random(tuple(s_.shape)) * 0.5 + 0.05).astype(np.float16)).to(s_.device))
        return model

    @staticmethod
    def pin_key(shape_name: str, K: int, cb: int, bsz: int, ctx: int | None = None) -> str:
        return f"{shape_name}:K{K}:cb{cb}:bs{bsz}" + (f":ctx{ctx}" if ctx is not None else "")

    # ---- checkpoints ----------------------------------------------------------------
    _HF = {"q": "self_attn.q_proj", "k": "self_attn.k_proj", "v": "self_attn.v_proj", "o": "self_attn.o_proj",
           "gate": "mlp.gate_proj", "up": "mlp.up_proj", "down": "mlp.down_proj"}

    def save_checkpoint(self, directory: str) -> None:
        """Write this (TP = 1) model as an EXL3 checkpoint directory with the reference's tensor naming for Llama (HF module names +
        .trellis/.suh/.svh/.mul1|.mcg, modules/linear.py:389-425) and a config.json with the fields from_checkpoint reads."""
        import json
        from safetensors.torch import save_file
        assert self.tp == 1, "save_checkpoint writes whole tensors: build the model with TP = 1"
        os.makedirs(directory, exist_ok=True)
        t = {}

        def put(key, lin):
            t[key + ".trellis"] = lin.trellis.cpu(); t[key + ".suh"] = lin.suh.cpu(); t[key + ".svh"] = lin.svh.cpu()
            if lin.mul1: t[key + ".mul1"] = torch.tensor([0xCAF6A435 - (1 << 32)], dtype=torch.int32)     # quantize.py:1414-1424 (value never read)
            if lin.mcg: t[key + ".mcg"] = torch.tensor([0xCBAC1FED - (1 << 32)], dtype=torch.int32)
            if lin.bias is not None: t[key + ".bias"] = lin.bias.cpu()

        for i, L in enumerate(self.layers):
            for nm, hf in self._HF.items():
                put(f"model.layers.{i}.{hf}", L[nm])
            t[f"model.layers.{i}.input_layernorm.weight"] = L["norm1"].cpu()
            t[f"model.layers.{i}.post_attention_layernorm.weight"] = L["norm2"].cpu()
        t["model.norm.weight"] = self.final_norm.cpu()
        put("lm_head", self.lm_head)
        save_file({k: v.contiguous() for k, v in t.items()}, os.path.join(directory, "model.safetensors"))
        s = self.shape
        cfg = {"architectures": ["LlamaForCausalLM"], "hidden_size": s.hidden, "intermediate_size": s.inter, "num_hidden_layers": self.n_layers,
               "num_attention_heads": s.heads_q, "num_key_value_heads": s.heads_kv, "head_dim": s.head_dim, "vocab_size": s.vocab,
               "rope_theta": s.rope_theta, "rms_norm_eps": self.eps, "quantization_config": {"quant_method": "exl3"}}
        json.dump(cfg, open(os.path.join(directory, "config.json"), "w"), indent=1)

    @classmethod
    def from_checkpoint(cls, directory: str, device: torch.device | str = "cuda:0", backend: TPBackendRCCL | None = None,
                        kv_bits: int = 4, max_ctx: int = 4096, layers: int | None = None) -> "SyntheticEXL3Llama":
        """The same hot path over a real EXL3 Llama checkpoint directory (config.json + *.safetensors).  Every tensor-parallel rank reads
        its own shards straight from the files (loader.load_linear_exl3 tp_slice): q/k/v/gate/up/lm_head column shards, o/down row shards
        (architecture/llama.py + modules/quant/exl3.py:284-330 split the same way).  Bitrates and codebooks may differ per tensor: a q|k|v or gate|up
        group of one kind runs as one fused launch, a mixed group as one GEMV per matrix (_same_kind)."""
        import json
        from . import loader
        cfg = json.load(open(os.path.join(directory, "config.json")))
        hq, hkv = cfg["num_attention_heads"], cfg.get("num_key_value_heads", cfg["num_attention_heads"])
        hd = cfg.get("head_dim") or cfg["hidden_size"] // hq
        shape = LlamaShape(os.path.basename(os.path.normpath(directory)), cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"],
                           hq, hkv, hd, cfg["vocab_size"], float(cfg.get("rope_theta", 10000.0)))
        stc = loader.SafetensorsCollection(directory)
        self = cls.__new__(cls)
        self.shape, self.kv_bits = shape, kv_bits
        if self.fx_act_in_gemv is None:
            self.fx_act_in_gemv = shape.hidden <= 2048
        self.device = torch.device(device)
        self.backend = backend or TPBackendRCCL(0, 1, self.device)
        self.tp, self.rank = self.backend.world_size, self.backend.rank
        tp, rank = self.tp, self.rank
        self.n_layers = layers or shape.layers
        assert shape.heads_kv % tp == 0, "TP degree must divide the KV heads (attention splits whole KV-head groups)"
        self.hq, self.hkv = shape.heads_q // tp, shape.heads_kv // tp
        ipts = split_points(shape.inter, tp)
        self.inter_local = ipts[rank + 1] - ipts[rank]
        vocab_pad = (shape.vocab + 127) // 128 * 128                        # stored padded to the Hadamard block (modules/linear.py:69-72)
        vpts = split_points(vocab_pad, tp)
        self.vocab_local = vpts[rank + 1] - vpts[rank]
        self.vocab_ldims = [vpts[r + 1] - vpts[r] for r in range(tp)]
        q0, q1 = rank * self.hq * hd, (rank + 1) * self.hq * hd
        kv0, kv1 = rank * self.hkv * hd, (rank + 1) * self.hkv * hd
        assert q0 % 128 == 0 and kv0 % 128 == 0 and (q1 - q0) % 128 == 0 and (kv1 - kv0) % 128 == 0, "head shards must be whole Hadamard blocks"
        whole = tp == 1

        def lin(key, sl, out_dtype=None):
            return loader.load_linear_exl3(stc, key, self.device, out_dtype, None if whole else sl)

        self.layers = []
        for i in range(self.n_layers):
            p = f"model.layers.{i}."
            L = {
                "q": lin(p + "self_attn.q_proj", (q0, q1, "n")), "k": lin(p + "self_attn.k_proj", (kv0, kv1, "n")),
                "v": lin(p + "self_attn.v_proj", (kv0, kv1, "n")), "o": lin(p + "self_attn.o_proj", (q0, q1, "k"), torch.float),
                "gate": lin(p + "mlp.gate_proj", (ipts[rank], ipts[rank + 1], "n")), "up": lin(p + "mlp.up_proj", (ipts[rank], ipts[rank + 1], "n")),
                "down": lin(p + "mlp.down_proj", (ipts[rank], ipts[rank + 1], "k"), torch.float),
                "norm1": stc.get_tensor(p + "input_layernorm.weight", self.device).half(),
                "norm2": stc.get_tensor(p + "post_attention_layernorm.weight", self.device).half(),
            }
            self.layers.append(L)
        self.final_norm = stc.get_tensor("model.norm.weight", self.device).half()
        head_key = "lm_head" if loader.is_exl3_storage(stc, "lm_head") else None
        if head_key is None:
            raise NotImplementedError("unquantized / tied lm_head: only EXL3 heads are on this path")
        self.lm_head = lin(head_key, (vpts[rank], vpts[rank + 1], "n"))
        l0 = self.layers[0]["q"]
        self.K, self.cb = l0.K, (2 if l0.mul1 else (1 if l0.mcg else 0))
        self.inv_freq = rope_inv_freq(hd, shape.rope_theta, cfg.get("rope_scaling")).to(self.device)
        self.eps = float(cfg.get("rms_norm_eps", 1e-5))
        self.page, self.max_ctx, self._state_bsz = 256, max_ctx, None
        return self

    # ---- state ---------------------------------------------------------------------------------------
    def alloc_state(self, bsz: int, pos: int = 1000):
        dev, s = self.device, self.shape
        hd = s.head_dim
        kv_dim = self.hkv * hd
        G = kv_dim // 32
        pages_per_seq = self.max_ctx // self.page
        self.block_table = torch.arange(bsz * pages_per_seq, dtype=torch.int32, device=dev).view(bsz, pages_per_seq)
        self.cache_seqlens = torch.full((bsz,), pos, dtype=torch.int32, device=dev)
        self.positions = torch.full((bsz,), pos, dtype=torch.int32, device=dev)
        n_pages = bsz * pages_per_seq
        self.kcache = [(torch.zeros((n_pages, self.page, G * self.kv_bits), dtype=torch.int32, device=dev),
                        torch.zeros((n_pages, self.page, G), dtype=torch.half, device=dev)) for _ in range(self.n_layers)]
        self.vcache = [(torch.zeros((n_pages, self.page, G * self.kv_bits), dtype=torch.int32, device=dev),
                        torch.zeros((n_pages, self.page, G), dtype=torch.half, device=dev)) for _ in range(self.n_layers)]
        f16, f32 = torch.half, torch.float
        if getattr(self, "_host_rng", None) is not None:
            import numpy as np
            self.x = torch.from_numpy(self._host_rng.standard_normal((bsz, s.hidden)).astype(np.float16)).to(dev)
        else:
            self.x = torch.randn((bsz, s.hidden), device=dev).to(f16)        # residual stream (fp16)
        self.x0 = self.x.clone()
        self.xn = torch.empty((bsz, s.hidden), dtype=f16, device=dev)
        self.q = torch.empty((bsz, 1, self.hq, hd), dtype=f16, device=dev)
        self.k = torch.empty((bsz, 1, self.hkv, hd), dtype=f16, device=dev)
        self.v = torch.empty((bsz, 1, self.hkv, hd), dtype=f16, device=dev)
        self.o = torch.empty((bsz, s.hidden), dtype=f32, device=dev)
        self.g = torch.empty((bsz, self.inter_local), dtype=f16, device=dev)
        self.u = torch.empty((bsz, self.inter_local), dtype=f16, device=dev)
        self.a = torch.empty((bsz, self.inter_local), dtype=f16, device=dev)
        self.d = torch.empty((bsz, s.hidden), dtype=f32, device=dev)
        self.logits = torch.empty((bsz, self.vocab_local), dtype=f16, device=dev)
        # fused pipeline: pre-rotated inputs (+ per-128-block sums) of every linear
        nb_h, nb_i, nb_a = s.hidden // 128, self.inter_local // 128, (self.hq * hd) // 128
        self.xh3 = [torch.empty((bsz, s.hidden), dtype=f16, device=dev) for _ in range(3)]
        self.xs3 = [torch.empty((bsz, nb_h), dtype=f32, device=dev) for _ in range(3)]
        self.xh_d = torch.empty((bsz, self.inter_local), dtype=f16, device=dev)
        self.ss = torch.empty((bsz, nb_h), dtype=f32, device=dev)
        # decode_step_resid ping-pongs the residual stream and its per-block sums of squares between two buffers
        self.x2 = torch.empty((bsz, s.hidden), dtype=f16, device=dev)
        self.ss2 = torch.empty((bsz, nb_h), dtype=f32, device=dev)
        # decode attention straight from the quantized cache (optional, head_dim 128): output, lengths incl. the new token, split partials
        self.attn_pos = pos
        self.attn_out = torch.empty((bsz, self.hq, hd), dtype=f16, device=dev)
        self.attn_lens = torch.full((bsz,), pos + 1, dtype=torch.int32, device=dev)
        self.attn_ws = torch.empty((bsz * self.hq * ((pos + 32) // 32) * 132,), dtype=f32, device=dev)
        self.rope_sin = torch.empty((bsz, 64), dtype=f32, device=dev)
        self.rope_cos = torch.empty((bsz, 64), dtype=f32, device=dev)
        self.kv_slots = torch.empty((bsz,), dtype=torch.long, device=dev)
        self.xs_d = torch.empty((bsz, nb_i), dtype=f32, device=dev)
        # decode_step_fx: the residual stream as a 64-bit fixed-point accumulator (value * 2^32)
        self.R = torch.zeros((bsz, s.hidden), dtype=torch.int64, device=dev)
        self.GU = torch.zeros((2, bsz, self.inter_local), dtype=torch.int64, device=dev)     # gate / up accumulators (fx_gu_atomic)
        self._state_bsz = bsz
        self._pstep = None                                 # (the persistent step's plan holds the cache / state pointers)

    # ---- one decode step (bsz tokens, one per sequence) -----------------------------------------------
    def decode_step(self):
        """Graph-capturable: only kernel launches on the current stream (+ RCCL all-reduces when tp > 1)."""
        bsz = self._state_bsz
        hd = self.shape.head_dim
        be = self.backend
        x = self.x
        x.copy_(self.x0)                                   # keep the step idempotent across replays
        pending = None                                     # fp32 sublayer output waiting for the fused residual add
        for li, L in enumerate(self.layers):
            if pending is None:
                ext.rms_norm(x, L["norm1"], self.xn, self.eps)
            else:
                ext.rms_norm_res_in(pending, L["norm1"], self.xn, x, self.eps)          # x += pending ; xn = norm(x)
            q2, k2, v2 = self.q.view(bsz, -1), self.k.view(bsz, -1), self.v.view(bsz, -1)
            if _same_kind(L["q"], L["k"], L["v"]):
                ext.exl3_mgemm_bcast(self.xn, [L["q"].trellis, L["k"].trellis, L["v"].trellis], [q2, k2, v2],
                                     [L["q"].suh, L["k"].suh, L["v"].suh], [L["q"].svh, L["k"].svh, L["v"].svh],
                                     L["q"].mcg, L["q"].mul1)
            else:
                L["q"].bc.run(self.xn, q2); L["k"].bc.run(self.xn, k2); L["v"].bc.run(self.xn, v2)
            ext.rope(self.q, self.q, self.k, self.k, self.inv_freq, 0, self.positions, None, 2, 1.0)
            kc, ks = self.kcache[li]
            vc, vs = self.vcache[li]
            ext.quant_cache_paged(self.k.view(bsz, 1, -1), kc, ks, self.v.view(bsz, 1, -1), vc, vs,
                                  self.cache_seqlens, self.block_table, self.page, 1)
            # [attention core out of scope: attention output := q]
            L["o"].bc.run(q2, self.o)
            be.all_reduce(self.o)
            ext.rms_norm_res_in(self.o, L["norm2"], self.xn, x, self.eps)             # x += o ; xn = norm(x)
            if _same_kind(L["gate"], L["up"]):
                ext.exl3_mgemm_bcast(self.xn, [L["gate"].trellis, L["up"].trellis], [self.g, self.u],
                                     [L["gate"].suh, L["up"].suh], [L["gate"].svh, L["up"].svh], L["gate"].mcg, L["gate"].mul1)
            else:
                L["gate"].bc.run(self.xn, self.g); L["up"].bc.run(self.xn, self.u)
            ext.silu_mul(self.g, self.u, self.a)
            L["down"].bc.run(self.a, self.d)
            be.all_reduce(self.d)
            pending = self.d
        ext.rms_norm_res_in(pending, self.final_norm, self.xn, x, self.eps)
        self.lm_head.bc.run(self.xn, self.logits)
        return self.logits

    # ---- the step through the reference's RUNNER objects -------------------------------------------------------------------------
    def decode_step_bc_runners(self):
        """The decode step as the reference's own module stack runs it over this build's seam: per layer RMSNorm -> `BC_Attention.run` -> residual add -> RMSNorm ->
        `BC_GatedMLP.run_bszN` -> residual add (modules/transformer.py's block order over libtorch/attention.cpp:506, libtorch/mlp.cpp:93), final norm,
        `BC_LinearEXL3.run` for the head -- what a maintainer gets who installs the `exllamav3_ext` stub and changes nothing else.  The attention over the
        quantized cache is part of `BC_Attention.run` (needs `alloc_state`); graph-capturable."""
        bsz, h, dev = self._state_bsz, self.shape.hidden, self.device
        if getattr(self, "_bc_runners", None) is None or self._bc_runners[0] != bsz:
            hd = self.shape.head_dim
            y = torch.empty((bsz, 1, h), dtype=torch.half, device=dev); d = torch.empty((1, bsz, h), dtype=torch.half, device=dev)
            m8 = max(8, bsz)
            guh = torch.empty((2, m8, h), dtype=torch.half, device=dev); gu = torch.empty((2, m8, self.inter_local), dtype=torch.half, device=dev)
            a = torch.empty((1, m8, self.inter_local), dtype=torch.half, device=dev); dxh = torch.empty_like(a)
            runners = []
            for li, L in enumerate(self.layers):
                (kc, ks), (vc, vs) = self.kcache[li], self.vcache[li]
                attn = ext.BC_Attention(num_q_heads=self.hq, num_kv_heads=self.hkv, head_dim=hd, hidden_size=h, hidden_size_padded=h, page_size=self.page,
                                        q_proj=L["q"].bc, k_proj=L["k"].bc, v_proj=L["v"].bc, o_proj=L["o"].bc, norm_eps=self.eps, inv_freq=self.inv_freq, rope_style=2,
                                        attn_factor=1.0, quant_cache=True, cache_k=kc, cache_v=vc, cache_k_scales=ks, cache_v_scales=vs, xh=None, h32=None)
                mlp = ext.BC_GatedMLP(guh, gu, a, dxh, None, None, None, L["gate"].K, L["gate"].mcg, L["gate"].mul1, True, False, False,
                                      L["gate"].bc, L["up"].bc, L["down"].bc, 0.0)
                runners.append((attn, mlp))
            self._bc_runners = (bsz, runners, y, d)
        _, runners, y, d = self._bc_runners
        x = self.x
        x.copy_(self.x0)
        for (attn, mlp), L in zip(runners, self.layers):
            ext.rms_norm(x, L["norm1"], self.xn, self.eps)
            attn.run(bsz, 1, self.xn.view(bsz, 1, h), y, self.cache_seqlens, self.block_table, 0, self.positions, None, None)
            ext.add(x, y.view(bsz, h))
            ext.rms_norm(x, L["norm2"], self.xn, self.eps)
            mlp.run_bszN(self.xn.view(1, bsz, h), d)
            ext.add(x, d.view(bsz, h))
        ext.rms_norm(x, self.final_norm, self.xn, self.eps)
        self.lm_head.bc.run(self.xn, self.logits)
        return self.logits

    # ---- the same step with the fused pipeline: deferred-epilogue GEMVs + glue kernels (8 launches per layer) ----------
    #: prefill: rebuild the next Linears' fp16 W on a side stream while the current GEMM runs (linear.ReconstructAhead).  Measured on
    #: MI355X: 68.3k tok/s with it vs 70.6k inline -- the concurrent kernel slows the hipBLASLt GEMM and W is no longer Infinity-Cache-warm
    #: when the GEMM reads it -- so it stays off; kept as a tested option.
    reconstruct_ahead = False
    #: prefill_chunk also runs the attention core (dequant_cache_paged + causal attention over the pages); off = the benchmark's default scope
    prefill_attention = False

    #: include the decode attention over the quantized cache in decode_step_fused (bench.py --attention)
    with_attention = False
    #: fx pipeline, head_dim 128: the flash-decoding merge of the attention's context splits runs inside o_proj's launch (ext.exl3_gemv_ex_attm) instead of
    #: as its own launch -- same bits, one launch less per layer
    attn_merge_in_oproj = os.environ.get("EXL3_HIP_ATTN_MERGE_IN_OPROJ", "1") != "0"
    #: ... and the q|k|v epilogue (split-k reduce, output Hadamard, RoPE, 4-bit append of the new token) runs inside the attention's context-split launch
    #: (ext.attn_decode_qcache_split_qkv) instead of glue_qkv_rs: 6 launches per layer with attention, same bits
    #: the same two fusions at head_dim 64 (two query heads per Hadamard block; exl3_gemv_ex_attm and the matrix-pipe split kernel take them): bit-identical.
    #: The 16-split cap of the 16-lane statistics makes the context splits longer, which the half-wave-per-token kernel pays for (Llama-3.2-1B with
    #: attention 1468 -> 1377 tok/s) and the matrix-pipe kernel does not (round 4, same box: 1466 half-wave kernel + merge launch, 1503 matrix-pipe kernel +
    #: merge launch, 1530 + merge inside o_proj, 1565 + q|k|v epilogue inside the split launch).  None = where the matrix-pipe kernel applies (at most 4
    #: query heads per kv head, 4-bit cache); EXL3_HIP_ATTN_MERGE_IN_OPROJ_HD64 = 0 / 1 forces it
    attn_merge_in_oproj_hd64 = {"0": False, "1": True}.get(os.environ.get("EXL3_HIP_ATTN_MERGE_IN_OPROJ_HD64", ""), None)
    attn_qkv_in_split = os.environ.get("EXL3_HIP_ATTN_QKV_IN_SPLIT", "1") != "0"
    #: fx pipeline WITHOUT the attention core (the linears-only step: o_proj's input is the finished q): glue_qkv_rs's work -- reduce, output Hadamard, RoPE,
    #: 4-bit K / V append -- runs inside o_proj's launch (ext.exl3_gemv_ex_qkvm): 5 launches per layer instead of 6 (4 with fx_act_in_gemv), same bits
    qkv_in_oproj = os.environ.get("EXL3_HIP_QKV_IN_OPROJ", "1") != "0"

    #: lm_head at m <= 4: glue_rotate + pre-rotated GEMV instead of the in-GEMV RMSNorm (the 1002-column-block launch repeats the 32 input
    #: Hadamards in every workgroup in NORM mode)
    rotate_for_head = os.environ.get("EXL3_HIP_ROTATE_FOR_HEAD", "1") != "0"       # measured +0.5 % at bs 1 (508.1 -> 510.5 tok/s)

    #: m <= 4: finish silu(g) * u inside the down GEMV instead of a glue_act launch
    act_in_gemv = os.environ.get("EXL3_HIP_ACT_IN_GEMV", "1") != "0"
    #: decode_step_fx runs up to this batch (the kernels take 4): measured 8B 4 bpw, same box: bs 1 +7..10 %, bs 2 +3 %, bs 4 -6.5 % against the glue pipeline
    #: (the atomics and the in-launch RMSNorm tasks scale with the rows)
    fx_max_bsz = 2
    #: decode_step_fx: silu(g) * u inside the down launch (5 launches per layer) instead of glue_act_rs + rotated-input down (6)
    #: None = by size (resolved in __init__): on for hidden <= 2048, where a launch is worth more than the redundant slab reads (Llama-3.2-1B:
    #: 1800 -> 1917 tok/s); off for Llama-3.1-8B (+0.2 .. +1.5 % tok/s over six same-box alternations, but the down launch then carries glue_act's
    #: work: 0.33 -> 0.30 of the per-launch HBM roofline).  EXL3_HIP_FX_ACT_IN_GEMV=0/1 overrides.
    fx_act_in_gemv = {"0": False, "1": True}.get(os.environ.get("EXL3_HIP_FX_ACT_IN_GEMV", ""), None)
    #: decode_step_fx: gate|up add into fixed-point accumulators, down forms silu(g) * u from them (5 launches per layer, no slab reduction)
    fx_gu_atomic = os.environ.get("EXL3_HIP_FX_GU_ATOMIC", "0") != "0"

    #: forced split-k factor per call type of the fused pipelines (0 = library heuristic); tools/sweep_split.py tunes these
    split = {"qkv": 0, "o": 0, "gu": 0, "down": 0}

    #: TP with the IPC all-reduce enabled: row-sharded linears hand their deferred slabs straight to the all-reduce launch (False: dense fp32 partials)
    ar_from_slabs = os.environ.get("EXL3_HIP_AR_FROM_SLABS", "1") == "1"

    #: per-step sin / cos / cache-row tables for glue_qkv (ext.qkv_prep: one launch per step; False = every layer's glue_qkv computes them itself)
    use_qkv_tab = True
    #: fx pipeline: the step's set-up pair (fx_init, qkv_prep) and its closing pair (fx_finish, glue_rotate for the lm_head) as one launch each
    fx_merged_boundaries = os.environ.get("EXL3_HIP_FX_MERGED_BOUNDARIES", "1") != "0"

    def _qkv_tab(self):
        """Launches ext.qkv_prep for this step's positions / block table and returns glue_qkv's `tab` argument (None when switched off)."""
        if not self.use_qkv_tab:
            return None
        ext.qkv_prep(self.inv_freq, self.positions, self.shape.head_dim, self.block_table, self.page, self.rope_sin, self.rope_cos, self.kv_slots)
        return (self.rope_sin, self.rope_cos, self.kv_slots)

    #: batches above 4 rows, one rank: glue_resid also rotates the new residual for its consumers (ext.glue_resid_rotate: 8 launches per layer
    #: instead of 10); False = separate glue_resid + glue_rotate launches (the reference's rounding point of the normalised activation)
    fold_rotate = True

    def _decode_step_fused_folded(self):
        """decode_step_fused for batches above 4 rows on one rank with the rotation folded into the residual kernel: per layer
        q|k|v (rotated input) -> glue_qkv_rs -> o -> glue_resid_rotate [-> xh for gate|up] -> gate|up -> glue_act_rs -> down ->
        glue_resid_rotate [-> xh for the next layer's q|k|v].  A folded launch normalises with the PREVIOUS residual's 1/rms (the new one needs
        the whole row); the consumers' epilogues multiply by r_new / r_prev.  First and last norm of the step use the separate kernels."""
        sp, bsz, hd = self.split, self._state_bsz, self.shape.head_dim
        ROT, DEF = ext.GEMV_IN_ROTATED, ext.GEMV_OUT_DEFERRED
        x = self.x
        x.copy_(self.x0)
        q2 = self.q.view(bsz, -1)
        hidden = self.shape.hidden
        ss_c, ss_o = self.ss, self.ss2                                    # sums of squares of the current residual / the other buffer
        tab = self._qkv_tab()
        ext.glue_resid(None, 0, None, None, x, ss_c, bsz)
        L0 = self.layers[0]
        ext.glue_rotate(x, ss_c, L0["norm1"], self.eps, [L0["q"].suh, L0["k"].suh, L0["v"].suh], self.xh3, bsz, xsums=self.xs3)
        rs = None                                                          # (ss_prev, ss_new) of the pending rescale, None = normalised exactly
        for li, L in enumerate(self.layers):
            lq, lk, lv, lo, lg, lu, ld = L["q"], L["k"], L["v"], L["o"], L["gate"], L["up"], L["down"]
            kc, ks = self.kcache[li]
            vc, vs = self.vcache[li]
            slabs, S = ext.exl3_gemv_ex(None, self.xh3, self.xs3, [lq.trellis, lk.trellis, lv.trellis], None, None, None,
                                        bsz, lq.mcg, lq.mul1, ROT | DEF, sp["qkv"])
            ext.glue_qkv_rs(slabs, S, lq.svh, lk.svh, lv.svh, self.q, None, None, self.inv_freq, self.positions, kc, ks, vc, vs,
                            self.block_table, self.page, self.kv_bits, self.kv_bits, bsz, self.hq, self.hkv, hd,
                            rs[0] if rs else None, rs[1] if rs else None, hidden, self.eps, tab=tab)
            o_in = q2
            if self.with_attention and hd in (64, 128):
                ext.attn_decode_qcache(self.q.view(bsz, self.hq, hd), self.attn_out, kc, ks, vc, vs, self.block_table, self.attn_lens,
                                       self.attn_pos + 1, workspace=self.attn_ws)
                o_in = self.attn_out.view(bsz, -1)
            so, So = ext.exl3_gemv_ex(o_in, None, None, [lo.trellis], None, [lo.suh], None, bsz, lo.mcg, lo.mul1, DEF, sp["o"])
            ext.glue_resid_rotate(so[0], So, lo.svh, None, x, ss_c, ss_o, L["norm2"], self.eps, [lg.suh, lu.suh], self.xh3[:2], bsz,
                                  xsums=self.xs3[:2])
            rs = (ss_c, ss_o); ss_c, ss_o = ss_o, ss_c
            sgu, Sgu = ext.exl3_gemv_ex(None, self.xh3[:2], self.xs3[:2], [lg.trellis, lu.trellis], None, None, None,
                                        bsz, lg.mcg, lg.mul1, ROT | DEF, sp["gu"])
            ext.glue_act_rs(sgu, Sgu, lg.svh, lu.svh, ld.suh, self.xh_d, self.xs_d, bsz, rs[0], rs[1], hidden, self.eps)
            sd, Sd = ext.exl3_gemv_ex(None, [self.xh_d], [self.xs_d], [ld.trellis], None, None, None, bsz, ld.mcg, ld.mul1, ROT | DEF, sp["down"])
            if li + 1 < self.n_layers:
                N = self.layers[li + 1]
                ext.glue_resid_rotate(sd[0], Sd, ld.svh, None, x, ss_c, ss_o, N["norm1"], self.eps, [N["q"].suh, N["k"].suh, N["v"].suh], self.xh3, bsz,
                                      xsums=self.xs3)
                rs = (ss_c, ss_o); ss_c, ss_o = ss_o, ss_c
            else:
                ext.glue_resid(sd[0], Sd, ld.svh, None, x, ss_c, bsz)
        ext.glue_rotate(x, ss_c, self.final_norm, self.eps, [self.lm_head.suh], self.xh3[:1], bsz, xsums=self.xs3[:1])
        ext.exl3_gemv_ex(None, self.xh3[:1], self.xs3[:1], [self.lm_head.trellis], [self.logits], None, [self.lm_head.svh],
                         bsz, self.lm_head.mcg, self.lm_head.mul1, ROT)
        return self.logits

    def decode_step_fused(self):
        """Fused decode step, 8 launches per layer: deferred-epilogue GEMVs + glue kernels.  The RMSNorm between two linears is
        split: glue_resid (distributed: split-k reduce + out-Hadamard + residual add + per-block sums of squares) and the consumer
        GEMV itself (GEMV_IN_NORM: normalise + input Hadamard while building its activation fragments)."""
        sp = self.split
        bsz = self._state_bsz
        hd = self.shape.head_dim
        be = self.backend
        x = self.x
        x.copy_(self.x0)
        ROT, DEF = ext.GEMV_IN_ROTATED, ext.GEMV_OUT_DEFERRED
        q2 = self.q.view(bsz, -1)
        ss = self.ss
        # batches above 4 rows: the norm + input Hadamards run once in glue_rotate instead of in every column-block workgroup
        rot = bsz > int(os.environ.get("EXL3_HIP_ROTATE_ABOVE", "4"))
        if rot and self.fold_rotate and self.tp == 1 and all(_same_kind(L["q"], L["k"], L["v"]) and _same_kind(L["gate"], L["up"]) for L in self.layers):
            return self._decode_step_fused_folded()
        tab = self._qkv_tab()
        # tensor parallel with the IPC all-reduce: o_proj / down_proj keep their deferred epilogue and the all-reduce launch finishes the slabs
        # (8 launches per layer instead of 10: no split-k reduce launch in front of either all-reduce)
        ar_slabs = self.tp > 1 and getattr(be, "ipc", None) is not None and self.ar_from_slabs and bsz * self.shape.hidden <= be.ipc.max_elems
        ext.glue_resid(None, 0, None, None, x, ss, bsz)
        for li, L in enumerate(self.layers):
            lq, lk, lv, lo, lg, lu, ld = L["q"], L["k"], L["v"], L["o"], L["gate"], L["up"], L["down"]
            kc, ks = self.kcache[li]
            vc, vs = self.vcache[li]
            if not _same_kind(lq, lk, lv):
                # q, k, v of different bitrates / codebooks: one GEMV each and the reference's standalone ops for this boundary
                ext.rms_norm(x, L["norm1"], self.xn, self.eps)
                lq.bc.run(self.xn, q2); lk.bc.run(self.xn, self.k.view(bsz, -1)); lv.bc.run(self.xn, self.v.view(bsz, -1))
                ext.rope(self.q, self.q, self.k, self.k, self.inv_freq, 0, self.positions, None, 2, 1.0)
                ext.quant_cache_paged(self.k.view(bsz, 1, -1), kc, ks, self.v.view(bsz, 1, -1), vc, vs, self.cache_seqlens, self.block_table, self.page, 1)
            else:
                if rot:
                    ext.glue_rotate(x, ss, L["norm1"], self.eps, [lq.suh, lk.suh, lv.suh], self.xh3, bsz, xsums=self.xs3)
                    slabs, S = ext.exl3_gemv_ex(None, self.xh3, self.xs3, [lq.trellis, lk.trellis, lv.trellis], None, None, None,
                                                bsz, lq.mcg, lq.mul1, ROT | DEF, sp["qkv"])
                else:
                    slabs, S = ext.exl3_gemv_ex_norm(x, L["norm1"], ss, self.eps, [lq.trellis, lk.trellis, lv.trellis], None, [lq.suh, lk.suh, lv.suh],
                                                     None, bsz, lq.mcg, lq.mul1, DEF, sp["qkv"])
                ext.glue_qkv(slabs, S, lq.svh, lk.svh, lv.svh, self.q, None, None, self.inv_freq, self.positions, kc, ks, vc, vs,
                             self.block_table, self.page, self.kv_bits, self.kv_bits, bsz, self.hq, self.hkv, hd, tab=tab)
            # attention core: out of the benchmark's scope by default (attention output := q); with_attention runs the
            # quant-cache-direct decode attention over the cached context (the K/V pages hold whatever the cache holds: zeros here except the
            # appended token, which is enough for timing and for the parity test that fills the cache first)
            o_in = q2
            if self.with_attention and hd in (64, 128):
                ext.attn_decode_qcache(self.q.view(bsz, self.hq, hd), self.attn_out, kc, ks, vc, vs, self.block_table, self.attn_lens,
                                       self.attn_pos + 1, workspace=self.attn_ws)
                o_in = self.attn_out.view(bsz, -1)
            if self.tp == 1:
                so, So = ext.exl3_gemv_ex(o_in, None, None, [lo.trellis], None, [lo.suh], None, bsz, lo.mcg, lo.mul1, DEF, sp["o"])
                ext.glue_resid(so[0], So, lo.svh, None, x, ss, bsz)
            elif ar_slabs:
                # row shard with a deferred epilogue: the IPC all-reduce launch finishes the slabs, reduces over the ranks and adds to the residual
                so, So = ext.exl3_gemv_ex(o_in, None, None, [lo.trellis], None, [lo.suh], None, bsz, lo.mcg, lo.mul1, DEF, sp["o"])
                be.all_reduce_resid_slabs(so[0], So, lo.svh, x, ss, bsz)
            else:
                lo.bc.run(o_in, self.o)
                be.all_reduce_resid(self.o, x, ss, bsz)                    # one-shot IPC push + residual add, or RCCL all-reduce + glue_resid
            if not _same_kind(lg, lu):
                # gate and up of different bitrates (at most one layer of a fractional-bpw checkpoint): one GEMV each + silu_mul, then the
                # down projection from the unrotated activation like o_proj
                ext.rms_norm(x, L["norm2"], self.xn, self.eps)
                lg.bc.run(self.xn, self.g); lu.bc.run(self.xn, self.u)
                ext.silu_mul(self.g, self.u, self.a)
                if self.tp == 1:
                    sd, Sd = ext.exl3_gemv_ex(self.a, None, None, [ld.trellis], None, [ld.suh], None, bsz, ld.mcg, ld.mul1, DEF, sp["down"])
                    ext.glue_resid(sd[0], Sd, ld.svh, None, x, ss, bsz)
                else:
                    ld.bc.run(self.a, self.d)
                    be.all_reduce_resid(self.d, x, ss, bsz)
                continue
            if rot:
                ext.glue_rotate(x, ss, L["norm2"], self.eps, [lg.suh, lu.suh], self.xh3[:2], bsz, xsums=self.xs3[:2])
                sgu, Sgu = ext.exl3_gemv_ex(None, self.xh3[:2], self.xs3[:2], [lg.trellis, lu.trellis], None, None, None,
                                            bsz, lg.mcg, lg.mul1, ROT | DEF, sp["gu"])
            else:
                sgu, Sgu = ext.exl3_gemv_ex_norm(x, L["norm2"], ss, self.eps, [lg.trellis, lu.trellis], None, [lg.suh, lu.suh], None,
                                                 bsz, lg.mcg, lg.mul1, DEF, sp["gu"])
            if self.tp == 1 and bsz == 1 and self.act_in_gemv:
                # silu(g) * u + input Hadamard inside the down GEMV (reads the gate / up slabs of the other workspace region): 7 launches / layer
                # (one rank only: at the TP = 2 shard shapes the ACT-mode launch measured 29 us per layer SLOWER than glue_act + rotated-input
                # GEMV, tools/ab_tp2.sh)
                sd, Sd = ext.exl3_gemv_ex_act(sgu, Sgu, lg.svh, lu.svh, ld.trellis, None, ld.suh, None, bsz, ld.mcg, ld.mul1, DEF, sp["down"])
                ext.glue_resid(sd[0], Sd, ld.svh, None, x, ss, bsz)
                continue
            ext.glue_act(sgu, Sgu, lg.svh, lu.svh, ld.suh, self.xh_d, self.xs_d, bsz)
            if self.tp == 1 or ar_slabs:
                sd, Sd = ext.exl3_gemv_ex(None, [self.xh_d], [self.xs_d], [ld.trellis], None, None, None, bsz, ld.mcg, ld.mul1, ROT | DEF, sp["down"])
                if self.tp == 1: ext.glue_resid(sd[0], Sd, ld.svh, None, x, ss, bsz)
                else: be.all_reduce_resid_slabs(sd[0], Sd, ld.svh, x, ss, bsz)
            else:
                ext.exl3_gemv_ex(None, [self.xh_d], [self.xs_d], [ld.trellis], [self.d], None, [ld.svh], bsz, ld.mcg, ld.mul1, ROT, c_fp32=True)
                be.all_reduce_resid(self.d, x, ss, bsz)
        if rot or self.rotate_for_head:
            ext.glue_rotate(x, ss, self.final_norm, self.eps, [self.lm_head.suh], self.xh3[:1], bsz, xsums=self.xs3[:1])
            ext.exl3_gemv_ex(None, self.xh3[:1], self.xs3[:1], [self.lm_head.trellis], [self.logits], None, [self.lm_head.svh],
                             bsz, self.lm_head.mcg, self.lm_head.mul1, ROT)
        else:
            ext.exl3_gemv_ex_norm(x, self.final_norm, ss, self.eps, [self.lm_head.trellis], [self.logits], [self.lm_head.suh], [self.lm_head.svh],
                                  bsz, self.lm_head.mcg, self.lm_head.mul1, 0)
        return self.logits

    def decode_step_resid(self):
        """Decode step with the residual adds folded into the consumer GEMVs: 5 launches per layer at batch <= 4, TP = 1
        (q|k|v [resid-in + norm], glue_qkv, o, gate|up [resid-in + norm], down [act-in]).  A GEMV_IN_RESID launch rebuilds
        resid += linear_out of the PREVIOUS linear from its split-k slabs for the Hadamard blocks of its own k-slice, normalises with the previous
        residual's 1/rms and lets the consumers of its outputs apply r_new / r_prev (the quantized linear commutes with the row scalar).  Other
        configurations (TP, batch > 4, mixed-kind groups) take decode_step_fused."""
        bsz = self._state_bsz
        same = all(_same_kind(L["q"], L["k"], L["v"]) and _same_kind(L["gate"], L["up"]) for L in self.layers)
        if self.tp != 1 or bsz > 4 or not same:
            return self.decode_step_fused()
        sp, hd, hidden = self.split, self.shape.head_dim, self.shape.hidden
        DEF = ext.GEMV_OUT_DEFERRED
        xc, xa, sc, sa = self.x, self.x2, self.ss, self.ss2               # current / alternate residual + sums of squares
        xc.copy_(self.x0)
        q2 = self.q.view(bsz, -1)
        tab = self._qkv_tab()
        ext.glue_resid(None, 0, None, None, xc, sc, bsz)                  # sums of squares of the embedding row(s)
        pend = None                                                        # (slab, S, svh) of the down_proj whose output is not yet in the residual
        # (column blocks per workgroup, k-slices) of the wave-per-column-block launches; tools/sweep_resid.py tunes these
        # o_proj / down_proj (32 column blocks) stay in the classic layout (cpw 0: the waves of a workgroup split the k-slice)
        wq, wo, wg, wd = sp.get("qkv_resid", (4, 16)), sp.get("o_resid", (0, 0)), sp.get("gu_resid", (14, 16)), sp.get("down_resid", (0, 0))
        for li, L in enumerate(self.layers):
            lq, lk, lv, lo, lg, lu, ld = L["q"], L["k"], L["v"], L["o"], L["gate"], L["up"], L["down"]
            kc, ks = self.kcache[li]
            vc, vs = self.vcache[li]
            if pend is None:
                slabs, S = ext.exl3_gemv_ex_norm(xc, L["norm1"], sc, self.eps, [lq.trellis, lk.trellis, lv.trellis], None, [lq.suh, lk.suh, lv.suh],
                                                 None, bsz, lq.mcg, lq.mul1, DEF, sp["qkv"])
                ext.glue_qkv(slabs, S, lq.svh, lk.svh, lv.svh, self.q, None, None, self.inv_freq, self.positions, kc, ks, vc, vs,
                             self.block_table, self.page, self.kv_bits, self.kv_bits, bsz, self.hq, self.hkv, hd, tab=tab)
            else:
                slabs, S = ext.exl3_gemv_ex_resid(xc, L["norm1"], sc, self.eps, pend[0], pend[1], pend[2], xa, sa,
                                                  [lq.trellis, lk.trellis, lv.trellis], [lq.suh, lk.suh, lv.suh], bsz, lq.mcg, lq.mul1, wq[1], wq[0])
                ext.glue_qkv_rs(slabs, S, lq.svh, lk.svh, lv.svh, self.q, None, None, self.inv_freq, self.positions, kc, ks, vc, vs,
                                self.block_table, self.page, self.kv_bits, self.kv_bits, bsz, self.hq, self.hkv, hd, sc, sa, hidden, self.eps, tab=tab)
                xc, xa, sc, sa = xa, xc, sa, sc
            o_in = q2
            if self.with_attention and hd in (64, 128):
                ext.attn_decode_qcache(self.q.view(bsz, self.hq, hd), self.attn_out, kc, ks, vc, vs, self.block_table, self.attn_lens,
                                       self.attn_pos + 1, workspace=self.attn_ws)
                o_in = self.attn_out.view(bsz, -1)
            if wo[0] > 0:
                so, So = ext.exl3_gemv_ex_wpc(o_in, [lo.trellis], [lo.suh], bsz, lo.mcg, lo.mul1, wo[0], wo[1])
            else:
                so, So = ext.exl3_gemv_ex(o_in, None, None, [lo.trellis], None, [lo.suh], None, bsz, lo.mcg, lo.mul1, DEF, wo[1])
            sgu, Sgu = ext.exl3_gemv_ex_resid(xc, L["norm2"], sc, self.eps, so[0], So, lo.svh, xa, sa, [lg.trellis, lu.trellis], [lg.suh, lu.suh],
                                              bsz, lg.mcg, lg.mul1, wg[1], wg[0])
            sd, Sd = ext.exl3_gemv_ex_act_rs(sgu, Sgu, lg.svh, lu.svh, sc, sa, hidden, self.eps, ld.trellis, None, ld.suh, None, bsz, ld.mcg, ld.mul1,
                                             DEF, wd[1], cpw=wd[0])
            xc, xa, sc, sa = xa, xc, sa, sc
            pend = (sd[0], Sd, ld.svh)
        ext.glue_resid(pend[0], pend[1], pend[2], None, xc, sc, bsz)
        self.x_final = xc
        if self.rotate_for_head:
            ext.glue_rotate(xc, sc, self.final_norm, self.eps, [self.lm_head.suh], self.xh3[:1], bsz, xsums=self.xs3[:1])
            ext.exl3_gemv_ex(None, self.xh3[:1], self.xs3[:1], [self.lm_head.trellis], [self.logits], None, [self.lm_head.svh],
                             bsz, self.lm_head.mcg, self.lm_head.mul1, ext.GEMV_IN_ROTATED)
        else:
            ext.exl3_gemv_ex_norm(xc, self.final_norm, sc, self.eps, [self.lm_head.trellis], [self.logits], [self.lm_head.suh], [self.lm_head.svh],
                                  bsz, self.lm_head.mcg, self.lm_head.mul1, 0)
        return self.logits

    def decode_step_fx(self):
        """Decode step with the residual stream in a fixed-point accumulator (round 3): 6 launches per layer (5 with fx_act_in_gemv, the default of small models) at batch <= 4, TP = 1 --
        q|k|v [reads R, RMSNorm inside], glue_qkv_rs, o_proj [adds into R], gate|up [reads R], glue_act_rs, down_proj [adds into R].
        o_proj / down_proj finish their own outputs (the output Hadamard is linear, so every split-k workgroup applies it and svh to its partial
        rows) and ADD them into R with integer atomics (order-independent: bit-reproducible); the two split-k-reduce + residual launches of the glue
        pipeline (glue_resid, 4.7 us each for < 100 KB) disappear.  A consumer normalises with the PREVIOUS residual's 1/rms (the sums of squares of
        the residual it reads are only complete once it has read every block: its column-block-0 workgroups publish them) and glue_qkv_rs /
        glue_act_rs multiply by r_new / r_prev -- decode_step_resid's protocol without its redundant slab reduction.  The fp16 residual of the
        reference (rounded after every add, norm.cu:193-218) is kept at higher precision here.  Other configurations take decode_step_fused."""
        bsz = self._state_bsz
        same = all(_same_kind(L["q"], L["k"], L["v"]) and _same_kind(L["gate"], L["up"]) for L in self.layers)
        if bsz > self.fx_max_bsz or not same or (self.tp != 1 and not self.fx_under_tp):
            return self.decode_step_fused()
        if self.tp != 1:
            return self._decode_step_fx_tp()
        sp, hd, hidden = dict(self.split), self.shape.head_dim, self.shape.hidden
        if sp["o"] == 0 and self.hq * hd == 4096 and hidden == 4096:
            sp["o"] = 8          # atomic epilogue: half the atomics of the dispatcher's 16-way split measured -0.8 us per layer (tools/sweep_split.py, STEP=decode_step_fx)
        ROT, DEF, ATOM = ext.GEMV_IN_ROTATED, ext.GEMV_OUT_DEFERRED, ext.GEMV_OUT_ATOMIC
        R = self.R
        sc, so_ = self.ss, self.ss2                                       # sums of squares: current (complete) / the buffer the next reader fills
        q2 = self.q.view(bsz, -1)
        if self.use_qkv_tab and self.fx_merged_boundaries:
            # the step's two set-up launches (fixed-point copy of the input rows, rope tables + cache rows) as one
            ext.fx_init_prep(self.x0, R, sc, bsz, self.inv_freq, self.positions, hd, self.block_table, self.page, self.rope_sin, self.rope_cos, self.kv_slots)
            tab = (self.rope_sin, self.rope_cos, self.kv_slots)
        else:
            tab = self._qkv_tab()
            ext.fx_init(self.x0, R, sc, bsz)
        for li, L in enumerate(self.layers):
            lq, lk, lv, lo, lg, lu, ld = L["q"], L["k"], L["v"], L["o"], L["gate"], L["up"], L["down"]
            kc, ks = self.kcache[li]
            vc, vs = self.vcache[li]
            slabs, S = ext.exl3_gemv_ex_fx(R, L["norm1"], sc, so_, self.eps, [lq.trellis, lk.trellis, lv.trellis], [lq.suh, lk.suh, lv.suh],
                                           bsz, lq.mcg, lq.mul1, sp["qkv"])
            attm = None                                                    # (partial records, splits): the merge runs inside o_proj's launch
            merge64 = self.attn_merge_in_oproj_hd64 if self.attn_merge_in_oproj_hd64 is not None else (self.hq // self.hkv <= 4 and self.kv_bits == 4)
            fuse_qkv = (self.with_attention and (hd == 128 or (hd == 64 and merge64)) and self.attn_merge_in_oproj and not self.fx_gu_atomic and self.attn_qkv_in_split
                        and tab is not None and self.kv_bits == 4)
            if fuse_qkv:
                # q|k|v epilogue (reduce, Hadamard, rope, cache append) inside the attention's context-split launch: 6 launches per layer with attention
                ns_, _ = ext.attn_decode_qcache_split_qkv(slabs, S, lq.svh, lk.svh, lv.svh, self.q.view(bsz, self.hq, hd), self.inv_freq, self.positions,
                                                          kc, ks, vc, vs, self.block_table, self.attn_lens, self.attn_pos + 1, self.attn_ws, tab,
                                                          sc, so_, hidden, self.eps)
                attm = (self.attn_ws, ns_)
            qkvm = (not self.with_attention and self.qkv_in_oproj and not self.fx_gu_atomic and tab is not None and self.kv_bits == 4 and hd in (64, 128))
            if not fuse_qkv and not qkvm:
                ext.glue_qkv_rs(slabs, S, lq.svh, lk.svh, lv.svh, self.q, None, None, self.inv_freq, self.positions, kc, ks, vc, vs,
                                self.block_table, self.page, self.kv_bits, self.kv_bits, bsz, self.hq, self.hkv, hd, sc, so_, hidden, self.eps, tab=tab)
            qk_sc, qk_so = sc, so_                                        # the q|k|v launch's (previous, new) sums of squares: the row-scale correction of its outputs
            sc, so_ = so_, sc
            o_in = q2
            if self.with_attention and hd in (64, 128) and not fuse_qkv:
                if self.attn_merge_in_oproj and not self.fx_gu_atomic and (hd == 128 or merge64):
                    attm = (self.attn_ws, ext.attn_decode_qcache_split(self.q.view(bsz, self.hq, hd), kc, ks, vc, vs, self.block_table, self.attn_lens,
                                                                       self.attn_pos + 1, self.attn_ws))
                else:
                    ext.attn_decode_qcache(self.q.view(bsz, self.hq, hd), self.attn_out, kc, ks, vc, vs, self.block_table, self.attn_lens,
                                           self.attn_pos + 1, workspace=self.attn_ws)
                    o_in = self.attn_out.view(bsz, -1)
            if self.fx_gu_atomic:
                # 5 launches per layer: gate|up ADD their rows into two fixed-point accumulators (cleared by the o_proj launch as a side job) and
                # down_proj forms silu(g) * u from them in its prologue -- no glue_act launch, no slab reduction anywhere in the MLP
                ext.fx_zero_next(self.GU)
                ext.exl3_gemv_ex(o_in, None, None, [lo.trellis], [R], [lo.suh], [lo.svh], bsz, lo.mcg, lo.mul1, ATOM, sp["o"])
                ext.exl3_gemv_ex_fx_atomic(R, L["norm2"], sc, so_, self.eps, [lg.trellis, lu.trellis], [self.GU[0], self.GU[1]], [lg.suh, lu.suh],
                                           [lg.svh, lu.svh], bsz, lg.mcg, lg.mul1, sp["gu"])
                ext.exl3_gemv_ex_actfx(self.GU[0], self.GU[1], sc, so_, hidden, self.eps, ld.trellis, R, ld.suh, ld.svh, bsz, ld.mcg, ld.mul1,
                                       ATOM, sp["down"])
                sc, so_ = so_, sc
                continue
            if attm is not None:
                ext.exl3_gemv_ex_attm(attm[0], attm[1], self.hq, self.hkv, lo.trellis, R, lo.suh, lo.svh, bsz, lo.mcg, lo.mul1, ATOM, sp["o"], head_dim=hd)
            elif qkvm:
                # no attention core: q|k|v epilogue (rope, K / V append) inside o_proj's launch, q finished by the preparation task that needs it
                ext.exl3_gemv_ex_qkvm(slabs, S, lq.svh, lk.svh, lv.svh, tab, qk_sc, qk_so, hidden, self.eps, hd, self.hkv, kc, ks, vc, vs,
                                      lo.trellis, R, lo.suh, lo.svh, bsz, lo.mcg, lo.mul1, ATOM, sp["o"], q_out=self.q)
            else:
                ext.exl3_gemv_ex(o_in, None, None, [lo.trellis], [R], [lo.suh], [lo.svh], bsz, lo.mcg, lo.mul1, ATOM, sp["o"])
            sgu, Sgu = ext.exl3_gemv_ex_fx(R, L["norm2"], sc, so_, self.eps, [lg.trellis, lu.trellis], [lg.suh, lu.suh], bsz, lg.mcg, lg.mul1, sp["gu"])
            if self.fx_act_in_gemv:
                # silu(g) * u (with the row-scale correction) + input Hadamard inside the down launch: 5 launches per layer
                ext.exl3_gemv_ex_act_rs(sgu, Sgu, lg.svh, lu.svh, sc, so_, hidden, self.eps, ld.trellis, R, ld.suh, ld.svh, bsz, ld.mcg, ld.mul1,
                                        ATOM, sp["down"])
                sc, so_ = so_, sc
                continue
            ext.glue_act_rs(sgu, Sgu, lg.svh, lu.svh, ld.suh, self.xh_d, self.xs_d, bsz, sc, so_, hidden, self.eps)
            sc, so_ = so_, sc
            ext.exl3_gemv_ex(None, [self.xh_d], [self.xs_d], [ld.trellis], [R], None, [ld.svh], bsz, ld.mcg, ld.mul1, ROT | ATOM, sp["down"])
        self.x_final = self.x
        if self.rotate_for_head and self.fx_merged_boundaries:
            # fp16 residual (kept for callers that read it), final norm and the lm_head's input rotation in one launch
            ext.fx_finish_rotate(R, self.x, sc, self.final_norm, self.eps, self.lm_head.suh, self.xh3[0], self.xs3[0], bsz)
            ext.exl3_gemv_ex(None, self.xh3[:1], self.xs3[:1], [self.lm_head.trellis], [self.logits], None, [self.lm_head.svh],
                             bsz, self.lm_head.mcg, self.lm_head.mul1, ROT)
        elif self.rotate_for_head:
            ext.fx_finish(R, self.x, sc, bsz)
            ext.glue_rotate(self.x, sc, self.final_norm, self.eps, [self.lm_head.suh], self.xh3[:1], bsz, xsums=self.xs3[:1])
            ext.exl3_gemv_ex(None, self.xh3[:1], self.xs3[:1], [self.lm_head.trellis], [self.logits], None, [self.lm_head.svh],
                             bsz, self.lm_head.mcg, self.lm_head.mul1, ROT)
        else:
            ext.fx_finish(R, self.x, sc, bsz)                             # fp16 residual + its sums of squares for the final norm
            ext.exl3_gemv_ex_norm(self.x, self.final_norm, sc, self.eps, [self.lm_head.trellis], [self.logits], [self.lm_head.suh], [self.lm_head.svh],
                                  bsz, self.lm_head.mcg, self.lm_head.mul1, 0)
        return self.logits

    #: the persistent decode step (round 5, ext.PersistentStep): None = wherever it applies -- with the direct residual edges and three decode-ahead units it is
    #: the faster step at both measured widths (tools/experiments/pstep_harness, same box: Llama-3.2-1B 2360 vs 1965 tok/s, Llama-3.1-8B 680 vs 641);
    #: EXL3_HIP_PSTEP=0/1 forces it.  Batch 1, one rank, mul1 codebook, 4-bit cache, no attention core (the linears-only step), one K for all layers and the head
    persistent = {"0": False, "1": True}.get(os.environ.get("EXL3_HIP_PSTEP", ""), None)

    def persistent_applies(self) -> bool:
        s = self.shape
        # one codebook; the tensors of a fused linear share their width; the layers' linears: one width or two adjacent ones (round 6); the lm_head: the layers' or 6 bits
        q0 = self.layers[0]["q"]
        lins = [L[n] for L in self.layers for n in ("q", "k", "v", "o", "gate", "up", "down")] + [self.lm_head]
        same = all(l.mul1 == q0.mul1 and l.mcg == q0.mcg for l in lins)
        same = same and all(_same_kind(L["q"], L["k"], L["v"]) and _same_kind(L["gate"], L["up"]) for L in self.layers)
        ks = sorted({l.K for l in lins[:-1]})
        same = same and len(ks) <= 2 and ks[-1] - ks[0] <= 1 and (self.lm_head.K == 6 or (len(ks) == 1 and self.lm_head.K == ks[0]))
        # with the attention core: inside the step (exl3_pstep.cuh: PS_ATTN) for at most 8 query heads per 128-value kv block
        att_ok = (not self.with_attention) or ((self.hq // self.hkv) * (128 // s.head_dim) <= 8 and os.environ.get("EXL3_HIP_PSTEP_ATTN", "1") != "0")
        # one rank -- or ONE rank's compute leg of a tensor-parallel job with the exchange left out (tp.OneRankOfMany: bench.py's llama-3.1-70b_tp8_rank line; a real TP
        # rank needs its peers' partial lines pushed into the step's slab buffers: not built)
        from .tp import OneRankOfMany
        import torch.distributed as dist
        # ... or a real tensor-parallel rank (round 6): the all-reduce behind o_proj / down_proj happens on the step's row edges (every rank pushes its partial lines into every
        # rank's exchange buffer over IPC-mapped addresses: ext.PersistentStep(tp=...)); EXL3_HIP_PSTEP_TP=0 keeps the launch-per-op TP branch
        one_rank = (self.tp == 1 or isinstance(self.backend, OneRankOfMany)
                    or (self.tp <= 8 and dist.is_available() and dist.is_initialized() and os.environ.get("EXL3_HIP_PSTEP_TP", "1") != "0"))
        return (self._state_bsz == 1 and one_rank and att_ok and self.kv_bits == 4 and same
                and s.hidden % 128 == 0 and s.hidden <= 8192 and s.head_dim in (64, 128) and self.use_qkv_tab)

    def decode_step_persistent(self):
        """The whole decode step as ONE launch (ext.PersistentStep / exl3_pstep.hip) behind the step's set-up launch (fx_init_prep: fixed-point copy of the
        input row, rope tables, cache rows): every quantized linear of every layer + the lm_head, RMSNorm, q|k|v epilogue with RoPE and the 4-bit K / V
        append, silu * mul and the residual adds -- and, with `with_attention`, the decode attention over the quantized cache inside o_proj's preparation.
        Same arithmetic per linear as decode_step_fx (generation 4's work unit, the same glue device functions);
        the residual is kept in fp32 between the linears; an RMSNorm's input is formed with the scale of the row's previous version and the linear's partial sums
        are corrected by (true scale / that scale) (decode_step_fx: 64-bit fixed point; the same correction for batches above 4).
        Falls back to decode_step_fx where it does not apply."""
        if not self.persistent_applies() or getattr(self, "_pstep_failed", False):
            return self.decode_step_fx()
        hd = self.shape.head_dim
        att = bool(self.with_attention)
        if getattr(self, "_pstep", None) is None or getattr(self, "_pstep_att", None) != att:
            layers = [dict(L, kcache=self.kcache[i], vcache=self.vcache[i]) for i, L in enumerate(self.layers)]
            from .tp import OneRankOfMany
            tp_real = self.tp > 1 and not isinstance(self.backend, OneRankOfMany)
            err_txt = None
            try:
                self._pstep = ext.PersistentStep(layers, self.lm_head, self.final_norm, self.shape.hidden, self.hq, self.hkv, hd, self.eps, rope_mode=2,
                                                 stamps=bool(os.environ.get("EXL3_HIP_PSTEP_STAMPS")), attention=att, tp=(self.tp, self.rank) if tp_real else None)
            except RuntimeError as e:
                err_txt = str(e)
                self._pstep = None
            if tp_real:
                # collective set-up: every rank takes part whatever failed locally -- the exchange buffers' IPC handles over the process group, the peers mapped, the table
                # committed, a barrier before anyone pushes; one rank without a plan -> every rank takes the launch-per-op TP branch
                import torch.distributed as dist
                handles = [None] * self.tp
                dist.all_gather_object(handles, self._pstep.tp_handle() if self._pstep is not None else None)
                ok = all(h is not None for h in handles)
                if ok:
                    try:
                        for r in range(self.tp):
                            if r != self.rank: self._pstep.tp_open_peer(r, handles[r])
                        self._pstep.tp_commit()
                    except RuntimeError as e:
                        ok, err_txt = False, str(e)
                flags = [None] * self.tp
                dist.all_gather_object(flags, ok)                       # (also the barrier: nobody pushes before every mapping exists)
                if not all(flags):
                    err_txt = err_txt or "a peer rank has no persistent plan"
                    self._pstep = None
            if self._pstep is None:
                e = err_txt
                # the planner / the device refused (no plan for this shape on this chip, the kernel does not fit a CU, out of memory for the repacked copy): the
                # launch-per-op pipeline computes the same step (persistent_applies() does not mirror every constraint of exl3_pstep_create)
                import warnings
                warnings.warn(f"exllamav3_amd: no persistent decode step for this model ({e}); using the launch-per-op pipeline", RuntimeWarning)
                self.persistent = False
                self._pstep = None
                self._pstep_failed = True                              # (no second attempt: under TP every attempt is a collective)
                return self.decode_step_fx()
            self._pstep_att = att
            self._pstep_checked = False
        elif self._pstep.tp is None and self._pstep.error_peek() and not torch.cuda.is_current_stream_capturing():
            # some earlier (possibly graph-replayed) step reported a timed-out wait to the pinned host word -- its logits were NaN; no synchronisation was needed to see it
            import warnings
            if self._pstep.error():
                warnings.warn("exllamav3_amd: the persistent decode step timed out (its logits were NaN-poisoned); using the launch-per-op pipeline from here on", RuntimeWarning)
                type(self).persistent = False
                self.persistent = False
                self._pstep = None
                return self.decode_step_fx()
        ext.fx_init_prep(self.x0, self.R, self.ss, 1, self.inv_freq, self.positions, hd, self.block_table, self.page, self.rope_sin, self.rope_cos, self.kv_slots)
        if att:
            self._pstep.run(self.R, self.logits, self.q, self.rope_sin, self.rope_cos, self.kv_slots, self.block_table, self.attn_lens, self.page)
        else:
            self._pstep.run(self.R, self.logits, self.q, self.rope_sin, self.rope_cos, self.kv_slots)
        if not self._pstep_checked and not torch.cuda.is_current_stream_capturing():
            # first run of a new plan: the step needs the whole grid co-resident (one workgroup per CU); if this device does not give that, its bounded waits time
            # out (flagged) -- then the launch-per-op pipeline takes over for good, loudly
            self._pstep_checked = True
            bad = bool(self._pstep.error())
            if self._pstep.tp is not None:
                # a tensor-parallel plan: the ranks decide TOGETHER (a rank that saw no time-out itself must not keep pushing into peers that stopped listening)
                import torch.distributed as dist
                votes = [None] * self.tp
                dist.all_gather_object(votes, bad)
                bad = any(votes)
            if bad:
                import warnings
                self._pstep_failed = True
                warnings.warn("exllamav3_amd: the persistent decode step timed out on this device (grid not co-resident?); using the launch-per-op pipeline", RuntimeWarning)
                type(self).persistent = False
                self.persistent = False
                self._pstep = None
                return self.decode_step_fx()
        return self.logits

    def decode_step_auto(self):
        """decode_step_persistent where it applies and is the faster step (see `persistent`), else decode_step_fx."""
        use = self.persistent if self.persistent is not None else True
        return self.decode_step_persistent() if (use and self.persistent_applies()) else self.decode_step_fx()

    #: tensor-parallel ranks take the fx pipeline too (round 4): 7 launches per layer instead of the glue pipeline's 8-10
    fx_under_tp = os.environ.get("EXL3_HIP_FX_UNDER_TP", "1") != "0"
    #: ... with silu(g) * u formed inside the down shard's launch (7 launches per layer) instead of glue_act_rs + rotated-input down (8)
    fx_tp_act_in_gemv = os.environ.get("EXL3_HIP_FX_TP_ACT_IN_GEMV", "0") != "0"

    def _decode_step_fx_tp(self):
        """decode_step_fx on a tensor-parallel rank (batch <= fx_max_bsz): the residual stream is the same 64-bit fixed-point accumulator R, replicated
        on every rank.  q|k|v and gate|up (column shards) read it exactly as on one rank; o_proj and down_proj (row shards) leave deferred slabs and ONE
        launch per boundary finishes them, exchanges the ranks' partial rows and adds the rank-order sum into R (IPC push: exl3_ar_reduce_fx; otherwise
        fp32 partial rows -> collective all-reduce -> exl3_fx_add).  The sums are formed in the same order on every rank and integer adds are exact, so
        R stays bit-identical across ranks.  7 launches per layer: q|k|v, glue_qkv_rs, o, all-reduce, gate|up, down (silu * mul inside), all-reduce."""
        bsz, hd, hidden, be = self._state_bsz, self.shape.head_dim, self.shape.hidden, self.backend
        sp = dict(self.split)
        DEF = ext.GEMV_OUT_DEFERRED
        R = self.R
        sc, so_ = self.ss, self.ss2
        q2 = self.q.view(bsz, -1)
        ext.fx_init_prep(self.x0, R, sc, bsz, self.inv_freq, self.positions, hd, self.block_table, self.page, self.rope_sin, self.rope_cos, self.kv_slots)
        tab = (self.rope_sin, self.rope_cos, self.kv_slots)
        slab_route = (getattr(be, "ipc", None) is not None and bsz * hidden <= be.ipc.max_elems) or getattr(be, "ipc_like", False)
        for li, L in enumerate(self.layers):
            lq, lk, lv, lo, lg, lu, ld = L["q"], L["k"], L["v"], L["o"], L["gate"], L["up"], L["down"]
            kc, ks = self.kcache[li]
            vc, vs = self.vcache[li]
            slabs, S = ext.exl3_gemv_ex_fx(R, L["norm1"], sc, so_, self.eps, [lq.trellis, lk.trellis, lv.trellis], [lq.suh, lk.suh, lv.suh],
                                           bsz, lq.mcg, lq.mul1, sp["qkv"])
            ext.glue_qkv_rs(slabs, S, lq.svh, lk.svh, lv.svh, self.q, None, None, self.inv_freq, self.positions, kc, ks, vc, vs,
                            self.block_table, self.page, self.kv_bits, self.kv_bits, bsz, self.hq, self.hkv, hd, sc, so_, hidden, self.eps, tab=tab)
            sc, so_ = so_, sc
            o_in = q2
            if self.with_attention and hd in (64, 128):
                ext.attn_decode_qcache(self.q.view(bsz, self.hq, hd), self.attn_out, kc, ks, vc, vs, self.block_table, self.attn_lens,
                                       self.attn_pos + 1, workspace=self.attn_ws)
                o_in = self.attn_out.view(bsz, -1)
            if slab_route:
                so, So = ext.exl3_gemv_ex(o_in, None, None, [lo.trellis], None, [lo.suh], None, bsz, lo.mcg, lo.mul1, DEF, sp["o"])
                be.all_reduce_fx_slabs(so[0], So, lo.svh, R, bsz)
            else:
                lo.bc.run(o_in, self.o)
                be.all_reduce_fx(self.o, R, bsz)
            sgu, Sgu = ext.exl3_gemv_ex_fx(R, L["norm2"], sc, so_, self.eps, [lg.trellis, lu.trellis], [lg.suh, lu.suh], bsz, lg.mcg, lg.mul1, sp["gu"])
            if not self.fx_tp_act_in_gemv:
                # silu(g) * u + down's input rotation as their own launch (the shard's k is short: few workgroups would repeat the slab reduction)
                ext.glue_act_rs(sgu, Sgu, lg.svh, lu.svh, ld.suh, self.xh_d, self.xs_d, bsz, sc, so_, hidden, self.eps)
                if slab_route:
                    sd, Sd = ext.exl3_gemv_ex(None, [self.xh_d], [self.xs_d], [ld.trellis], None, None, None, bsz, ld.mcg, ld.mul1, ext.GEMV_IN_ROTATED | DEF, sp["down"])
                    be.all_reduce_fx_slabs(sd[0], Sd, ld.svh, R, bsz)
                else:
                    ext.exl3_gemv_ex(None, [self.xh_d], [self.xs_d], [ld.trellis], [self.d], None, [ld.svh], bsz, ld.mcg, ld.mul1, ext.GEMV_IN_ROTATED, c_fp32=True)
                    be.all_reduce_fx(self.d, R, bsz)
            elif slab_route:
                sd, Sd = ext.exl3_gemv_ex_act_rs(sgu, Sgu, lg.svh, lu.svh, sc, so_, hidden, self.eps, ld.trellis, None, ld.suh, None, bsz, ld.mcg, ld.mul1,
                                                 DEF, sp["down"])
                be.all_reduce_fx_slabs(sd[0], Sd, ld.svh, R, bsz)
            else:
                ext.exl3_gemv_ex_act_rs(sgu, Sgu, lg.svh, lu.svh, sc, so_, hidden, self.eps, ld.trellis, self.d, ld.suh, ld.svh, bsz, ld.mcg, ld.mul1,
                                        0, sp["down"], c_fp32=True)
                be.all_reduce_fx(self.d, R, bsz)
            sc, so_ = so_, sc
        self.x_final = self.x
        ext.fx_finish_rotate(R, self.x, sc, self.final_norm, self.eps, self.lm_head.suh, self.xh3[0], self.xs3[0], bsz)
        ext.exl3_gemv_ex(None, self.xh3[:1], self.xs3[:1], [self.lm_head.trellis], [self.logits], None, [self.lm_head.svh],
                         bsz, self.lm_head.mcg, self.lm_head.mul1, ext.GEMV_IN_ROTATED)
        return self.logits

    def decode_step_fused_v1(self):
        """First-generation glue pipeline: RMSNorm + input Hadamards in a single-workgroup glue_norm launch (kept as the
        comparison baseline for decode_step_fused; same bits)."""
        sp = self.split
        bsz = self._state_bsz
        hd = self.shape.head_dim
        be = self.backend
        x = self.x
        x.copy_(self.x0)
        ROT, DEF = ext.GEMV_IN_ROTATED, ext.GEMV_OUT_DEFERRED
        q2 = self.q.view(bsz, -1)
        pend = None                                            # (slab ptr, S, svh) of the pending down_proj, or a dense fp32 tensor
        for li, L in enumerate(self.layers):
            lq, lk, lv, lo, lg, lu, ld = L["q"], L["k"], L["v"], L["o"], L["gate"], L["up"], L["down"]
            tq = ([lq.suh, lk.suh, lv.suh], self.xh3, self.xs3)
            if pend is None:
                ext.glue_norm(None, 0, None, None, x, L["norm1"], self.eps, *tq, bsz)
            elif isinstance(pend, torch.Tensor):
                ext.glue_norm(None, 0, None, None, x, L["norm1"], self.eps, *tq, bsz, y_dense=pend)
            else:
                ext.glue_norm(pend[0], pend[1], pend[2], None, x, L["norm1"], self.eps, *tq, bsz)
            slabs, S = ext.exl3_gemv_ex(None, self.xh3, self.xs3, [lq.trellis, lk.trellis, lv.trellis], None, None, None,
                                        bsz, lq.mcg, lq.mul1, ROT | DEF, sp["qkv"])
            kc, ks = self.kcache[li]
            vc, vs = self.vcache[li]
            ext.glue_qkv(slabs, S, lq.svh, lk.svh, lv.svh, self.q, None, None, self.inv_freq, self.positions, kc, ks, vc, vs,
                         self.block_table, self.page, self.kv_bits, self.kv_bits, bsz, self.hq, self.hkv, hd)
            # [attention core out of scope: attention output := q]; o_proj takes the raw attention output (fused input Hadamard)
            t2 = ([lg.suh, lu.suh], self.xh3[:2], self.xs3[:2])
            if self.tp == 1:
                so, So = ext.exl3_gemv_ex(q2, None, None, [lo.trellis], None, [lo.suh], None, bsz, lo.mcg, lo.mul1, DEF, sp["o"])
                ext.glue_norm(so[0], So, lo.svh, None, x, L["norm2"], self.eps, *t2, bsz)
            else:
                lo.bc.run(q2, self.o)
                be.all_reduce(self.o)
                ext.glue_norm(None, 0, None, None, x, L["norm2"], self.eps, *t2, bsz, y_dense=self.o)
            sgu, Sgu = ext.exl3_gemv_ex(None, self.xh3[:2], self.xs3[:2], [lg.trellis, lu.trellis], None, None, None,
                                        bsz, lg.mcg, lg.mul1, ROT | DEF, sp["gu"])
            ext.glue_act(sgu, Sgu, lg.svh, lu.svh, ld.suh, self.xh_d, self.xs_d, bsz)
            if self.tp == 1:
                sd, Sd = ext.exl3_gemv_ex(None, [self.xh_d], [self.xs_d], [ld.trellis], None, None, None, bsz, ld.mcg, ld.mul1, ROT | DEF, sp["down"])
                pend = (sd[0], Sd, ld.svh)
            else:
                ext.exl3_gemv_ex(None, [self.xh_d], [self.xs_d], [ld.trellis], [self.d], None, [ld.svh], bsz, ld.mcg, ld.mul1, ROT, c_fp32=True)
                be.all_reduce(self.d)
                pend = self.d
        th = ([self.lm_head.suh], self.xh3[:1], self.xs3[:1])
        if isinstance(pend, torch.Tensor):
            ext.glue_norm(None, 0, None, None, x, self.final_norm, self.eps, *th, bsz, y_dense=pend)
        else:
            ext.glue_norm(pend[0], pend[1], pend[2], None, x, self.final_norm, self.eps, *th, bsz)
        ext.exl3_gemv_ex(None, self.xh3[:1], self.xs3[:1], [self.lm_head.trellis], [self.logits], None, [self.lm_head.svh],
                         bsz, self.lm_head.mcg, self.lm_head.mul1, ROT)
        return self.logits

    def decode_step_tail(self):
        """Decode step with in-kernel tail epilogues: 4 launches per layer (qkv+rope+KV-append, o+add+norm, gate/up+act, down+add+norm).
        Tensor-parallel ranks need the all-reduce between o/down and the norm, so TP > 1 (and head_dim 64) uses decode_step_fused."""
        if self.tp != 1 or self.shape.head_dim != 128:             # the q|k|v tail epilogue handles one head per Hadamard block only
            return self.decode_step_fused()
        bsz, hd = self._state_bsz, self.shape.head_dim
        x = self.x
        x.copy_(self.x0)
        q2 = self.q.view(bsz, -1)
        ext.rope_table(self.inv_freq, self.positions, self.rope_sin, self.rope_cos)
        L0 = self.layers[0]
        ext.glue_norm(None, 0, None, None, x, L0["norm1"], self.eps, [L0["q"].suh, L0["k"].suh, L0["v"].suh], self.xh3, self.xs3, bsz)
        for li, L in enumerate(self.layers):
            lq, lk, lv, lo, lg, lu, ld = L["q"], L["k"], L["v"], L["o"], L["gate"], L["up"], L["down"]
            kc, ks = self.kcache[li]
            vc, vs = self.vcache[li]
            ext.exl3_gemv_qkv(None, self.xh3, self.xs3, [lq.trellis, lk.trellis, lv.trellis], None, [lq.svh, lk.svh, lv.svh], bsz, lq.mcg, lq.mul1,
                              self.q, None, None, self.rope_sin, self.rope_cos, self.positions, kc, ks, vc, vs, self.block_table, self.page,
                              self.kv_bits, self.kv_bits, self.hq, self.hkv, hd)
            # [attention core out of scope: attention output := q]
            ext.exl3_gemv_norm(q2, None, None, lo.trellis, lo.suh, lo.svh, None, bsz, lo.mcg, lo.mul1, x, L["norm2"], self.eps,
                               [lg.suh, lu.suh], self.xh3[:2], self.xs3[:2])
            ext.exl3_gemv_act(None, self.xh3[:2], self.xs3[:2], [lg.trellis, lu.trellis], None, [lg.svh, lu.svh], bsz, lg.mcg, lg.mul1,
                              ld.suh, self.xh_d, self.xs_d)
            if li + 1 < len(self.layers):
                N = self.layers[li + 1]
                nw, tg, nx = N["norm1"], [N["q"].suh, N["k"].suh, N["v"].suh], 3
            else:
                nw, tg, nx = self.final_norm, [self.lm_head.suh], 1
            ext.exl3_gemv_norm(None, self.xh_d, self.xs_d, ld.trellis, None, ld.svh, None, bsz, ld.mcg, ld.mul1, x, nw, self.eps,
                               tg, self.xh3[:nx], self.xs3[:nx])
        ext.exl3_gemv_ex(None, self.xh3[:1], self.xs3[:1], [self.lm_head.trellis], [self.logits], None, [self.lm_head.svh],
                         bsz, self.lm_head.mcg, self.lm_head.mul1, ext.GEMV_IN_ROTATED)
        return self.logits

    def gemv_calls(self, pipeline):
        """Zero-argument callables, one per quantized-GEMV launch of a decode step (bench.py roofline leg).
        pipeline: "tail" | "glue" | "unfused" (True / False are accepted for glue / unfused)."""
        if pipeline is True: pipeline = "glue"
        if pipeline is False: pipeline = "unfused"
        if self.tp != 1 and pipeline == "tail": pipeline = "glue"
        if pipeline == "resid" and (self.tp != 1 or self._state_bsz > 4): pipeline = "glue"
        if pipeline == "fx" and (self.tp != 1 or self._state_bsz > self.fx_max_bsz): pipeline = "glue"
        bsz, hd = self._state_bsz, self.shape.head_dim
        q2, k2, v2 = self.q.view(bsz, -1), self.k.view(bsz, -1), self.v.view(bsz, -1)
        ROT, DEF = ext.GEMV_IN_ROTATED, ext.GEMV_OUT_DEFERRED
        calls = []
        for li, L in enumerate(self.layers):
            lq, lk, lv, lo, lg, lu, ld = L["q"], L["k"], L["v"], L["o"], L["gate"], L["up"], L["down"]
            if pipeline == "tail":
                kc, ks = self.kcache[li]
                vc, vs = self.vcache[li]
                calls.append(lambda lq=lq, lk=lk, lv=lv, kc=kc, ks=ks, vc=vc, vs=vs: ext.exl3_gemv_qkv(
                    None, self.xh3, self.xs3, [lq.trellis, lk.trellis, lv.trellis], None, [lq.svh, lk.svh, lv.svh], bsz, lq.mcg, lq.mul1,
                    self.q, None, None, self.rope_sin, self.rope_cos, self.positions, kc, ks, vc, vs, self.block_table, self.page,
                    self.kv_bits, self.kv_bits, self.hq, self.hkv, hd))
                calls.append(lambda lo=lo, lg=lg, lu=lu, L=L: ext.exl3_gemv_norm(
                    q2, None, None, lo.trellis, lo.suh, lo.svh, None, bsz, lo.mcg, lo.mul1, self.x, L["norm2"], self.eps,
                    [lg.suh, lu.suh], self.xh3[:2], self.xs3[:2]))
                calls.append(lambda lg=lg, lu=lu, ld=ld: ext.exl3_gemv_act(
                    None, self.xh3[:2], self.xs3[:2], [lg.trellis, lu.trellis], None, [lg.svh, lu.svh], bsz, lg.mcg, lg.mul1,
                    ld.suh, self.xh_d, self.xs_d))
                calls.append(lambda ld=ld, lq=lq, lk=lk, lv=lv, L=L: ext.exl3_gemv_norm(
                    None, self.xh_d, self.xs_d, ld.trellis, None, ld.svh, None, bsz, ld.mcg, ld.mul1, self.x, L["norm1"], self.eps,
                    [lq.suh, lk.suh, lv.suh], self.xh3, self.xs3))
            elif pipeline == "fx" and self.tp == 1 and bsz <= 4:
                # decode_step_fx's four GEMV launches per layer (the atomic ones keep adding into R: timing only)
                ATOM = ext.GEMV_OUT_ATOMIC
                calls.append(lambda lq=lq, lk=lk, lv=lv, L=L: ext.exl3_gemv_ex_fx(self.R, L["norm1"], self.ss, self.ss2, self.eps, [lq.trellis, lk.trellis, lv.trellis], [lq.suh, lk.suh, lv.suh], bsz, lq.mcg, lq.mul1))
                calls.append(lambda lo=lo: ext.exl3_gemv_ex(q2, None, None, [lo.trellis], [self.R], [lo.suh], [lo.svh], bsz, lo.mcg, lo.mul1, ATOM))
                calls.append(lambda lg=lg, lu=lu, L=L: ext.exl3_gemv_ex_fx(self.R, L["norm2"], self.ss, self.ss2, self.eps, [lg.trellis, lu.trellis], [lg.suh, lu.suh], bsz, lg.mcg, lg.mul1))
                if self.fx_act_in_gemv and not self.fx_gu_atomic:
                    # the down launch forms silu(g) * u from gate|up's slabs (those of one sample launch stand in for the chain: timing only)
                    if li == 0:
                        gu0 = ext.exl3_gemv_ex_fx(self.R, L["norm2"], self.ss, self.ss2, self.eps, [lg.trellis, lu.trellis], [lg.suh, lu.suh], bsz, lg.mcg, lg.mul1)
                    calls.append(lambda ld=ld, lg=lg, lu=lu: ext.exl3_gemv_ex_act_rs(gu0[0], gu0[1], lg.svh, lu.svh, self.ss, self.ss2, self.shape.hidden, self.eps,
                                                                                      ld.trellis, self.R, ld.suh, ld.svh, bsz, ld.mcg, ld.mul1, ATOM))
                else:
                    calls.append(lambda ld=ld: ext.exl3_gemv_ex(None, [self.xh_d], [self.xs_d], [ld.trellis], [self.R], None, [ld.svh], bsz, ld.mcg, ld.mul1, ROT | ATOM))
            elif pipeline == "resid" and self.tp == 1 and bsz <= 4:
                # decode_step_resid's four GEMV launches per layer; producer slabs of one sample launch each stand in for the chain (timing only)
                sp, hidden = self.split, self.shape.hidden
                wq, wo, wg, wd = sp.get("qkv_resid", (4, 16)), sp.get("o_resid", (0, 0)), sp.get("gu_resid", (14, 16)), sp.get("down_resid", (0, 0))
                if wo[0] > 0:
                    o_call = lambda lo=lo: ext.exl3_gemv_ex_wpc(q2, [lo.trellis], [lo.suh], bsz, lo.mcg, lo.mul1, wo[0], wo[1])
                else:
                    o_call = lambda lo=lo: ext.exl3_gemv_ex(q2, None, None, [lo.trellis], None, [lo.suh], None, bsz, lo.mcg, lo.mul1, DEF, wo[1])
                if li == 0:
                    so0 = o_call()
                    gu0 = ext.exl3_gemv_ex_resid(self.x, L["norm2"], self.ss, self.eps, so0[0][0], so0[1], lo.svh, self.x2, self.ss2,
                                                 [lg.trellis, lu.trellis], [lg.suh, lu.suh], bsz, lg.mcg, lg.mul1, wg[1], wg[0])
                    sd0 = ext.exl3_gemv_ex_act_rs(gu0[0], gu0[1], lg.svh, lu.svh, self.ss, self.ss2, hidden, self.eps, ld.trellis, None, ld.suh, None,
                                                  bsz, ld.mcg, ld.mul1, DEF, wd[1], cpw=wd[0])
                calls.append(lambda lq=lq, lk=lk, lv=lv, L=L, ld=ld: ext.exl3_gemv_ex_resid(
                    self.x, L["norm1"], self.ss, self.eps, sd0[0][0], sd0[1], ld.svh, self.x2, self.ss2, [lq.trellis, lk.trellis, lv.trellis],
                    [lq.suh, lk.suh, lv.suh], bsz, lq.mcg, lq.mul1, wq[1], wq[0]))
                calls.append(o_call)
                calls.append(lambda lg=lg, lu=lu, L=L, lo=lo: ext.exl3_gemv_ex_resid(
                    self.x, L["norm2"], self.ss, self.eps, so0[0][0], so0[1], lo.svh, self.x2, self.ss2, [lg.trellis, lu.trellis], [lg.suh, lu.suh],
                    bsz, lg.mcg, lg.mul1, wg[1], wg[0]))
                calls.append(lambda ld=ld, lg=lg, lu=lu: ext.exl3_gemv_ex_act_rs(
                    gu0[0], gu0[1], lg.svh, lu.svh, self.ss, self.ss2, hidden, self.eps, ld.trellis, None, ld.suh, None, bsz, ld.mcg, ld.mul1, DEF,
                    wd[1], cpw=wd[0]))
            elif pipeline == "glue" and self.tp == 1 and bsz > 4:
                # batches above 4 rows: the step rotates once (glue_rotate) and the GEMVs read pre-rotated inputs
                # (the step's own arguments: block sums for the mul1 raw variant, the forced splits of self.split)
                sp = self.split
                calls.append(lambda lq=lq, lk=lk, lv=lv: ext.exl3_gemv_ex(None, self.xh3, self.xs3, [lq.trellis, lk.trellis, lv.trellis], None, None, None, bsz, lq.mcg, lq.mul1, ROT | DEF, sp["qkv"]))
                calls.append(lambda lo=lo: ext.exl3_gemv_ex(q2, None, None, [lo.trellis], None, [lo.suh], None, bsz, lo.mcg, lo.mul1, DEF, sp["o"]))
                calls.append(lambda lg=lg, lu=lu: ext.exl3_gemv_ex(None, self.xh3[:2], self.xs3[:2], [lg.trellis, lu.trellis], None, None, None, bsz, lg.mcg, lg.mul1, ROT | DEF, sp["gu"]))
                calls.append(lambda ld=ld: ext.exl3_gemv_ex(None, [self.xh_d], [self.xs_d], [ld.trellis], None, None, None, bsz, ld.mcg, ld.mul1, ROT | DEF, sp["down"]))
            elif pipeline == "glue" and self.tp == 1:
                calls.append(lambda lq=lq, lk=lk, lv=lv, L=L: ext.exl3_gemv_ex_norm(self.x, L["norm1"], self.ss, self.eps, [lq.trellis, lk.trellis, lv.trellis], None, [lq.suh, lk.suh, lv.suh], None, bsz, lq.mcg, lq.mul1, DEF))
                calls.append(lambda lo=lo: ext.exl3_gemv_ex(q2, None, None, [lo.trellis], None, [lo.suh], None, bsz, lo.mcg, lo.mul1, DEF))
                calls.append(lambda lg=lg, lu=lu, L=L: ext.exl3_gemv_ex_norm(self.x, L["norm2"], self.ss, self.eps, [lg.trellis, lu.trellis], None, [lg.suh, lu.suh], None, bsz, lg.mcg, lg.mul1, DEF))
                if bsz == 1 and self.act_in_gemv:
                    if li == 0:
                        gu0 = ext.exl3_gemv_ex_norm(self.x, L["norm2"], self.ss, self.eps, [lg.trellis, lu.trellis], None, [lg.suh, lu.suh], None, bsz, lg.mcg, lg.mul1, DEF)
                    calls.append(lambda ld=ld, lg=lg, lu=lu, gu0=gu0: ext.exl3_gemv_ex_act(gu0[0], gu0[1], lg.svh, lu.svh, ld.trellis, None, ld.suh, None, bsz, ld.mcg, ld.mul1, DEF))
                else:
                    calls.append(lambda ld=ld: ext.exl3_gemv_ex(None, [self.xh_d], [self.xs_d], [ld.trellis], None, None, None, bsz, ld.mcg, ld.mul1, ROT | DEF))
            else:
                calls.append(lambda lq=lq, lk=lk, lv=lv: ext.exl3_mgemm_bcast(self.xn, [lq.trellis, lk.trellis, lv.trellis], [q2, k2, v2], [lq.suh, lk.suh, lv.suh], [lq.svh, lk.svh, lv.svh], lq.mcg, lq.mul1))
                calls.append(lambda lo=lo: lo.bc.run(q2, self.o))
                calls.append(lambda lg=lg, lu=lu: ext.exl3_mgemm_bcast(self.xn, [lg.trellis, lu.trellis], [self.g, self.u], [lg.suh, lu.suh], [lg.svh, lu.svh], lg.mcg, lg.mul1))
                calls.append(lambda ld=ld: ld.bc.run(self.a, self.d))
        if pipeline == "glue" and self.tp == 1 and bsz > 4:
            calls.append(lambda: ext.exl3_gemv_ex(None, self.xh3[:1], self.xs3[:1], [self.lm_head.trellis], [self.logits], None, [self.lm_head.svh], bsz, self.lm_head.mcg, self.lm_head.mul1, ROT))
        elif pipeline in ("glue", "fx") and self.tp == 1:
            calls.append(lambda: ext.exl3_gemv_ex_norm(self.x, self.final_norm, self.ss, self.eps, [self.lm_head.trellis], [self.logits], [self.lm_head.suh], [self.lm_head.svh], bsz, self.lm_head.mcg, self.lm_head.mul1, 0))
        elif pipeline == "tail" and self.tp == 1:
            calls.append(lambda: ext.exl3_gemv_ex(None, self.xh3[:1], self.xs3[:1], [self.lm_head.trellis], [self.logits], None, [self.lm_head.svh], bsz, self.lm_head.mcg, self.lm_head.mul1, ROT))
        else:
            calls.append(lambda: self.lm_head.bc.run(self.xn, self.logits))
        return calls

    def gemv_launches_per_step(self):
        """(k, n_total, count) of every quantized-GEMV launch of one decode step, for the roofline accounting."""
        s, hd = self.shape, self.shape.head_dim
        per_layer = [(s.hidden, (self.hq + 2 * self.hkv) * hd), (self.hq * hd, s.hidden),
                     (s.hidden, 2 * self.inter_local), (self.inter_local, s.hidden)]
        return [(k, n, self.n_layers) for (k, n) in per_layer] + [(s.hidden, self.vocab_local, 1)]

    def prefill_allreduce_dtype(self):
        """dtype of the (tokens, hidden) partial sums a TP rank hands to the all-reduce in prefill_chunk (o / down are created with fp32 outputs:
        architecture/llama.py:95,111)."""
        return self.layers[0]["o"].default_out_dtype

    # ---- one prefill chunk -----------------------------------------------------------------------------
    def prefill_chunk(self, tokens: int):
        """All linears + norms + rope + KV-quant of a `tokens`-token chunk (bsz 1); attention core out of scope.
        rows > 144 -> reconstruct_had_slice + hgemm (modules/quant/exl3.py:161-218)."""
        s, hd, dev = self.shape, self.shape.head_dim, self.device
        if getattr(self, "_pf_tokens", None) != tokens:
            self.px0 = torch.randn((tokens, s.hidden), device=dev).half()
            self._pf_tokens = tokens
            G = self.hkv * hd // 32
            pages = (tokens + self.page - 1) // self.page
            self.pf_bt = torch.arange(pages, dtype=torch.int32, device=dev).view(1, pages)
            self.pf_sl = torch.zeros((1,), dtype=torch.int32, device=dev)
            self.pf_cache = [torch.zeros((pages, self.page, G * self.kv_bits), dtype=torch.int32, device=dev) for _ in range(2)] + \
                            [torch.zeros((pages, self.page, G), dtype=torch.half, device=dev) for _ in range(2)]
            # prefill attention (optional): dequantized pages, lengths incl. the chunk, attention output
            self.pf_kd = torch.zeros((pages, self.page, self.hkv * hd), dtype=torch.half, device=dev)
            self.pf_vd = torch.zeros_like(self.pf_kd)
            self.pf_len = torch.full((1,), tokens, dtype=torch.int32, device=dev)
            self.pf_ao = torch.empty((tokens, self.hq * hd), dtype=torch.half, device=dev)
        x = self.px0.clone()
        be = self.backend
        xn = torch.empty_like(x)
        ahead = None
        if self.reconstruct_ahead and tokens >= 1024:
            from .linear import LinearEXL3, ReconstructAhead
            ahead = getattr(self, "_ahead", None)
            if ahead is None:
                seq = [L[nm] for L in self.layers for nm in ("q", "k", "v", "o", "gate", "up", "down")]
                ahead = self._ahead = ReconstructAhead(seq, depth=2)
            LinearEXL3.ahead = ahead
            ahead.begin()
        for L in self.layers:
            ext.rms_norm(x, L["norm1"], xn, self.eps)
            qkv = type(L["q"]).forward_multi([L["q"], L["k"], L["v"]], xn) if hd == 128 else None
            if qkv is not None:
                # one GEMM for q|k|v; rope and the KV-cache append read its column ranges in place
                q, k, v = qkv
                ext.rope_strided(q, k, self.inv_freq, 0, None, None, 1.0, 1, tokens)
                ext.quant_cache_paged_strided(k, self.pf_cache[0], self.pf_cache[2], v, self.pf_cache[1], self.pf_cache[3], self.pf_sl, self.pf_bt,
                                              self.page, tokens)
            else:
                q = L["q"].forward(xn).view(1, tokens, self.hq, hd)
                k = L["k"].forward(xn).view(1, tokens, self.hkv, hd)
                v = L["v"].forward(xn).view(1, tokens, self.hkv, hd)
                ext.rope(q, q, k, k, self.inv_freq, 0, None, None, 2, 1.0)
                ext.quant_cache_paged(k.view(1, tokens, -1), self.pf_cache[0], self.pf_cache[2], v.view(1, tokens, -1), self.pf_cache[1],
                                      self.pf_cache[3], self.pf_sl, self.pf_bt, self.page, tokens)
                q = q.view(tokens, -1)
            if self.prefill_attention:
                # the reference's prefill attention step: dequantize the (just appended) pages, causal attention of the chunk over them
                # (cache/quant.py:83-117 + flash_attn_with_kvcache; here exl3_attn_prefill.hip).  q may be a column range of the fused q|k|v output.
                ext.dequant_cache_paged(self.pf_cache[0], self.pf_cache[2], self.pf_kd, self.pf_cache[1], self.pf_cache[3], self.pf_vd, self.pf_len,
                                        self.pf_bt, self.page)
                q2 = q if q.dim() == 2 else q.reshape(tokens, -1)
                q4 = q2.as_strided((1, tokens, self.hq, hd), (tokens * q2.stride(0), q2.stride(0), hd, 1), q2.storage_offset())
                ext.attn_prefill_paged(q4, self.pf_ao.view(1, tokens, self.hq, hd), self.pf_kd.view(-1, self.page, self.hkv, hd),
                                       self.pf_vd.view(-1, self.page, self.hkv, hd), self.pf_bt, self.pf_len)
                q = self.pf_ao
            if self.tp == 1:
                L["o"].forward_add_residual(q, x)                           # residual add in the GEMM epilogue (q: attention output, or its stand-in)
                ext.rms_norm(x, L["norm2"], xn, self.eps)
            else:
                o = L["o"].forward(q.contiguous().view(tokens, -1))
                be.all_reduce(o)
                ext.rms_norm_res_in(o, L["norm2"], xn, x, self.eps)
            a = type(L["gate"]).forward_gate_up_silu(L["gate"], L["up"], xn)      # one GEMM for gate|up when the rows allow it
            if self.tp == 1:
                L["down"].forward_add_residual(a, x)
            else:
                d = L["down"].forward(a)
                be.all_reduce(d)
                ext.add(x, d)
        if ahead is not None:
            from .linear import LinearEXL3
            ahead.end()
            LinearEXL3.ahead = None
        self.px_out = x                                                     # residual stream after the last layer (parity tests read sampled rows)
        ext.rms_norm(x[-1:], self.final_norm, xn[-1:], self.eps)
        return self.lm_head.forward(xn[-1:].contiguous())
