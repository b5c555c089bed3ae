#!/usr/bin/env python3
"""
Generate tests/golden/ref_python.npz by IMPORTING the reference's own Python (read-only at
/root/reference) in the build container and running its pure-torch CPU code on seeded inputs.

The reference's native extension cannot be built here (CUDA/PTX only), so
`exllamav3.ext` is replaced by a stub before import: every function below that is exercised is
pure Python/torch.  Where the reference composition calls ext.reconstruct (get_weight_tensor),
the stub returns a caller-supplied W_hat, so what is pinned is the reference's COMPOSITION
(Hadamard side/order/scale, suh/svh placement), not our decode.

Run:  python tests/golden/make_golden.py        (only possible where /root/reference exists)
Nothing under tests/, bench.py or smoke() reads /root/reference at run time; they read the .npz.
"""
import sys, types, importlib, os, math
import numpy as np
import torch

REF = "/root/reference/exllamav3"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_python.npz")


def fake_pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


class ExtStub(types.ModuleType):
    what = {}

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        if k in self.what:
            return self.what[k]

        def f(*a, **kw):
            raise RuntimeError("stubbed ext." + k)
        return f


def main():
    fake_pkg("exllamav3", REF)
    stub = ExtStub("exllamav3_ext")
    extmod = types.ModuleType("exllamav3.ext")
    extmod.exllamav3_ext = stub
    sys.modules["exllamav3.ext"] = extmod
    fake_pkg("exllamav3.modules", REF + "/modules")
    fake_pkg("exllamav3.modules.quant", REF + "/modules/quant")
    fake_pkg("exllamav3.modules.quant.exl3_lib", REF + "/modules/quant/exl3_lib")

    quantize = importlib.import_module("exllamav3.modules.quant.exl3_lib.quantize")
    hadamard = importlib.import_module("exllamav3.util.hadamard")
    rope_mod = importlib.import_module("exllamav3.util.rope")

    out = {}
    g = torch.Generator().manual_seed(1234)

    # --- tile permutation (modules/quant/exl3_lib/quantize.py:21-49)
    out["perm"] = quantize.tensor_core_perm("cpu").numpy()
    out["perm_i"] = quantize.tensor_core_perm_i("cpu").numpy()
    out["codebook_scale"] = np.float64(quantize.codebook_scale)
    out["mcg_mult"] = np.uint64(quantize.codebook_mcg_mult)
    out["mul1_mult"] = np.uint64(quantize.codebook_mul1_mult)

    # --- Hadamard (util/hadamard.py:108-137, exl3_lib/quantize.py:340-357)
    out["had128"] = hadamard.get_hadamard_dt(128, "cpu", torch.float32, 1 / math.sqrt(128)).numpy()
    w = torch.randn(256, 384, generator=g, dtype=torch.float32)
    out["had_in"] = w.numpy()
    out["had_l"] = quantize.preapply_had_l(w.clone(), 128).numpy()
    out["had_r"] = quantize.preapply_had_r(w.clone(), 128).numpy()

    # --- LinearEXL3.get_weight_tensor composition + unpack_bf (modules/quant/exl3.py:142-158,221-237)
    # exl3.py imports model.config; build the class namespace without it by exec'ing only what we need
    # is fragile, so import the module with a minimal fake for ...model.config and ...util.tensor.
    fake_pkg("exllamav3.model", REF + "/model")
    cfgmod = types.ModuleType("exllamav3.model.config")
    class Config: pass
    class NullConfig:
        class infer_params: no_reconstruct = False
    cfgmod.Config, cfgmod.NullConfig = Config, NullConfig
    sys.modules["exllamav3.model.config"] = cfgmod
    exl3 = importlib.import_module("exllamav3.modules.quant.exl3")
    k, n = 256, 384
    w_hat = (torch.randn(k, n, generator=g) * 1.2).half()
    suh = (torch.sign(torch.randn(k, generator=g)) * (0.5 + torch.rand(k, generator=g))).half()
    svh = (torch.sign(torch.randn(n, generator=g)) * (0.5 + torch.rand(n, generator=g))).half()

    def fake_reconstruct(dst, trellis, K, mcg, mul1):
        dst.copy_(w_hat)
    stub.what["reconstruct"] = fake_reconstruct
    stub.what["BC_LinearEXL3"] = lambda *a, **kw: None
    obj = exl3.LinearEXL3.__new__(exl3.LinearEXL3)
    obj.su = None; obj.sv = None; obj.suh = suh; obj.svh = svh
    obj.trellis = torch.zeros((k // 16, n // 16, 16 * 4), dtype=torch.int16)
    obj.K = 4; obj.in_features = k; obj.out_features = n; obj.mcg = False; obj.mul1 = False
    obj.transformers_fix = False
    out["wt_w_hat"] = w_hat.numpy()
    out["wt_suh"] = suh.numpy()
    out["wt_svh"] = svh.numpy()
    out["wt_weight"] = obj.get_weight_tensor().float().numpy()
    bf = torch.randint(-32768, 32767, (24,), generator=g, dtype=torch.int32).to(torch.int16)
    out["bf_in"] = bf.numpy()
    out["bf_out"] = obj.unpack_bf(bf).numpy()

    # --- RoPE torch reference (util/rope.py:199-222 llama3 scaling, :340-352 sincos, :365-432 apply_torch)
    for name, style in (("neox", rope_mod.RopeStyle.NEOX), ("gptj", rope_mod.RopeStyle.GPTJ)):
        rs = rope_mod.RopeSettings(
            head_dim=128, rope_theta=500000.0,
            rope_scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0,
                          "high_freq_factor": 4.0, "original_max_position_embeddings": 8192},
            max_position_embeddings=131072, rope_style=style)
        r = rope_mod.RoPE("cpu", rs)
        q = torch.randn(2, 5, 4, 128, generator=g).half()
        kk = torch.randn(2, 5, 2, 128, generator=g).half()
        out[f"rope_{name}_inv_freq"] = r.inv_freq.float().numpy()
        out[f"rope_{name}_q"] = q.numpy()
        out[f"rope_{name}_k"] = kk.numpy()
        qo, ko = r.apply_torch(q.clone(), kk.clone(), pos=37)
        out[f"rope_{name}_pos37_q"] = qo.numpy(); out[f"rope_{name}_pos37_k"] = ko.numpy()
        positions = torch.tensor([3, 1000], dtype=torch.int)
        qo, ko = r.apply_torch(q.clone(), kk.clone(), positions=positions)
        out[f"rope_{name}_positions"] = positions.numpy()
        out[f"rope_{name}_posv_q"] = qo.numpy(); out[f"rope_{name}_posv_k"] = ko.numpy()
        pid = torch.randint(0, 5000, (2, 5), generator=g, dtype=torch.int)
        qo, ko = r.apply_torch(q.clone(), kk.clone(), position_ids=pid)
        out[f"rope_{name}_pid"] = pid.numpy()
        out[f"rope_{name}_pid_q"] = qo.numpy(); out[f"rope_{name}_pid_k"] = ko.numpy()
    rs64 = rope_mod.RopeSettings(head_dim=64, rope_theta=10000.0)
    r64 = rope_mod.RoPE("cpu", rs64)
    out["rope_default64_inv_freq"] = r64.inv_freq.float().numpy()

    # --- RMSNorm torch reference (modules/rmsnorm.py:65-79), called unbound on a plain namespace
    try:
        modbase = types.ModuleType("exllamav3.modules.module")
        class Module:
            def __init__(self, *a, **kw): pass
        modbase.Module = Module
        sys.modules["exllamav3.modules"].Module = Module
        tpalloc = types.ModuleType("exllamav3.model.model_tp_alloc")
        class TPAllocation: pass
        tpalloc.TPAllocation = TPAllocation
        sys.modules["exllamav3.model.model_tp_alloc"] = tpalloc
        rms = importlib.import_module("exllamav3.modules.rmsnorm")
        for tag, dim, rows, cb, cs in (("a", 4096, 3, 0.0, 1.0), ("b", 2048, 2, 1.0, 0.5), ("c", 128, 7, 0.0, 1.0)):
            x = (torch.randn(rows, dim, generator=g) * 2.0).half()
            wv = (1.0 + 0.1 * torch.randn(dim, generator=g)).half()
            ns = types.SimpleNamespace(rms_norm_eps=1e-5, constant_scale=cs, constant_bias=cb,
                                       unweighted=False, weight=wv, out_dtype=torch.half)
            y = rms.RMSNorm.forward_torch(ns, x, {}, torch.half)
            out[f"rms_{tag}_x"] = x.numpy(); out[f"rms_{tag}_w"] = wv.numpy()
            out[f"rms_{tag}_y"] = y.numpy()
            out[f"rms_{tag}_cfg"] = np.array([1e-5, cb, cs], dtype=np.float64)
    except Exception as e:          # pragma: no cover
        print("RMSNorm reference import failed:", repr(e))
        raise

    # --- MoE router, torch path of the reference (modules/block_sparse_mlp.py:95-127 routing_std_bias with bsz > 1: logits = y @ gate (+ bias),
    # top-k of the logits, weights = softmax over the selected logits in fp32 -> fp16; with no bias this is routing_std's contract, which the
    # reference itself only implements natively (routing.cu:457-590)).  The function is compiled from the reference file's own source text.
    import ast
    src = open(REF + "/modules/block_sparse_mlp.py").read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "routing_std_bias"][0]
    ns = {"torch": torch, "ext": stub}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), REF + "/modules/block_sparse_mlp.py", "exec"), ns)
    for tag, E, topk, with_bias in (("mixtral", 8, 2, False), ("wide", 64, 6, True)):
        y = torch.randn(5, 256, generator=g).half()
        gate = (torch.randn(256, E, generator=g) * 0.1).half()
        bias = (torch.randn(E, generator=g) * 0.5).half() if with_bias else None
        cfg = types.SimpleNamespace(gate_tensor=gate, gate_tensor_t=None, router_bias=bias, num_experts=E, num_experts_per_tok=topk,
                                    per_expert_scale=None)
        sel, wts = ns["routing_std_bias"](5, cfg, y, {})
        out[f"route_{tag}_y"] = y.numpy(); out[f"route_{tag}_gate"] = gate.numpy()
        out[f"route_{tag}_bias"] = bias.numpy() if with_bias else np.zeros(0, np.float16)
        out[f"route_{tag}_sel"] = sel.numpy(); out[f"route_{tag}_w"] = wts.numpy()

    # --- paged-cache attention semantics (modules/attention_fn/torch.py:81-160 `_torch_bighead_fallback`: the reference's own pure-torch
    # stand-in for flash_attn_with_kvcache -- new k / v appended at cache_seqlens, causal lower-right mask, GQA): pins the oracle's
    # attn_prefill / attn_decode_qcache.  fp32 tensors, so the values are the definition, not a kernel's rounding.
    fake_pkg("exllamav3.modules.attention_fn", REF + "/modules/attention_fn")
    attn_t = importlib.import_module("exllamav3.modules.attention_fn.torch")
    attn_t.has_warned_sdpa_fallback = True
    for tag, (bsz, q_len, hq, hkv, hd, page, pps, lens) in {
        "a": (2, 7, 4, 2, 64, 16, 4, [20, 0]),          # a chunk over a context / over an empty cache, crossing page edges
        "b": (3, 1, 8, 2, 32, 16, 3, [5, 31, 16]),      # decode: one new token per sequence
    }.items():
        npages = bsz * pps
        bt = torch.randperm(npages, generator=g).view(bsz, pps).to(torch.int32)
        kc = torch.randn(npages, page, hkv, hd, generator=g)
        vc = torch.randn(npages, page, hkv, hd, generator=g)
        q = torch.randn(bsz, q_len, hq, hd, generator=g)
        kn = torch.randn(bsz, q_len, hkv, hd, generator=g)
        vn = torch.randn(bsz, q_len, hkv, hd, generator=g)
        cl = torch.tensor(lens, dtype=torch.int32)
        out[f"attn_{tag}_q"] = q.numpy(); out[f"attn_{tag}_k_new"] = kn.numpy(); out[f"attn_{tag}_v_new"] = vn.numpy()
        out[f"attn_{tag}_k_cache"] = kc.numpy().copy(); out[f"attn_{tag}_v_cache"] = vc.numpy().copy()
        out[f"attn_{tag}_block_table"] = bt.numpy(); out[f"attn_{tag}_cache_seqlens"] = cl.numpy()
        o_ = attn_t._torch_bighead_fallback(q, kn, vn, kc, vc, bt.long(), cl, causal=True, softmax_scale=hd ** -0.5)
        out[f"attn_{tag}_out"] = o_.float().numpy()

    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
    main()
