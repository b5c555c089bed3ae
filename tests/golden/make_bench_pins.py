#!/usr/bin/env python3
"""Pinned logits for bench.py's correctness check (VERDICT r3 "next" 1 d).

For every configuration below, SyntheticEXL3Llama.pin_model builds ONE layer of the benchmark's shape + a 2048-column lm_head from a fixed numpy
PCG64 seed (host-generated: the same tensors on every machine), and THIS script runs the ORACLE (oracle/exl3_oracle.py, numpy, CPU) over it: the
composition of tests/test_gpu_path.py:_oracle_decode.  The logits are committed as tests/golden/bench_pins.json.

  * tests/test_bench_pins.py (CPU suite) re-derives every pin from the oracle and compares -> the file cannot drift from the oracle;
  * tests/test_gpu_bench.py (GPU suite) and bench.py run the shipped decode pipeline over the same model on the GPU and compare within the end-to-end
    tolerance 3e-2 * RMS (the bar of every pipeline-vs-oracle test).

Run from the repository root:  python tests/golden/make_bench_pins.py        (no GPU, ~1 minute on 8 cores)
"""
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import exl3_oracle as o                                     # noqa: E402

#: (shape name, K, codebook, batch): the decode configurations bench.py puts on the driver line for the Llama shapes
CONFIGS = [("llama-3.1-8b", 4, 2, 1), ("llama-3.1-8b", 4, 0, 1), ("llama-3.1-8b", 4, 2, 16), ("llama-3.2-1b", 4, 2, 1), ("llama-3.1-70b", 3, 2, 1),
           # (shape, K, codebook, batch, context): the ..._with_attention_ctx* lines -- the decode attention over a pre-filled 4-bit cache of `context` tokens in the step
           # (round 6; the 16 000-token pin is the regime where a context split of the persistent step's attention takes several 128-token steps)
           ("llama-3.1-8b", 4, 2, 1, 1000), ("llama-3.1-8b", 4, 2, 1, 16000), ("llama-3.2-1b", 4, 2, 1, 1000)]
PINS = os.path.join(ROOT, "tests", "golden", "bench_pins.json")
_POOL = ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 8))


def _np(t):
    return t.detach().cpu().numpy()


def _lin(L, x, out_fp32=False, chunk=512):
    """oracle linear over column chunks on a thread pool (numpy releases the GIL)."""
    tr, su, sv = _np(L.trellis), _np(L.suh), _np(L.svh)
    cb = 1 if L.mcg else (2 if L.mul1 else 0)
    n = sv.shape[0]

    def one(c0):
        t, s_u, s_v, _ = o.tp_slice(tr, su, sv, None, c0, min(c0 + chunk, n), "n")
        return o.linear_forward(x, t, s_u, s_v, L.K, cb, out_fp32=out_fp32)
    return np.concatenate(list(_POOL.map(one, range(0, n, chunk))), axis=-1)


def oracle_logits(model) -> np.ndarray:
    """The oracle composition of one decode step over `model` (attention core out of scope: attention output := rope(q), as in the benchmark; with
    model.with_attention: the new token's K / V rows quantized into the cache, attention over the dequantized cache -- tests/test_gpu_fullsize.py's
    _oracle_attention_sublayer, reference libtorch/attention.cpp:386-504 + cache/q_cache_kernels.cuh:160-236)."""
    s, x, pending = model.shape, _np(model.x0).copy(), None
    b = x.shape[0]
    att = bool(getattr(model, "with_attention", False))
    for li, L in enumerate(model.layers):
        if pending is None: xn = o.rms_norm(x, _np(L["norm1"]), model.eps)
        else: xn, x = o.rms_norm(pending, _np(L["norm1"]), model.eps, residual_in=x)
        q, k = _lin(L["q"], xn), _lin(L["k"], xn)
        q4, _ = o.rope(q.reshape(b, 1, model.hq, s.head_dim), k.reshape(b, 1, model.hkv, s.head_dim), _np(model.inv_freq),
                       positions=_np(model.positions), rope_mode=o.ROPE_NEOX)
        a_in = q4.reshape(b, -1)
        if att:
            assert b == 1
            pos, hd = int(model.positions[0]), s.head_dim
            v = _lin(L["v"], xn)
            _, k4 = o.rope(q.reshape(b, 1, model.hq, hd), k.reshape(b, 1, model.hkv, hd), _np(model.inv_freq), positions=_np(model.positions), rope_mode=o.ROPE_NEOX)
            (kc, ksc), (vc, vsc) = model.kcache[li], model.vcache[li]
            # (contiguous pages per sequence in these models: row t of the flat cache is token t)
            kq, ks = _np(kc).view(np.uint32).reshape(1, -1, kc.shape[-1]).copy(), _np(ksc).reshape(1, -1, ksc.shape[-1]).copy()
            vq, vs = _np(vc).view(np.uint32).reshape(1, -1, vc.shape[-1]).copy(), _np(vsc).reshape(1, -1, vsc.shape[-1]).copy()
            knq, kns = o.kv_quant(k4.reshape(b, 1, -1), 4); vnq, vns = o.kv_quant(v.reshape(b, 1, -1), 4)
            kq[:, pos:pos + 1] = knq; ks[:, pos:pos + 1] = kns; vq[:, pos:pos + 1] = vnq; vs[:, pos:pos + 1] = vns
            kd = o.kv_dequant(kq[:, :pos + 1], ks[:, :pos + 1], 4).reshape(b, -1, model.hkv, hd); vd = o.kv_dequant(vq[:, :pos + 1], vs[:, :pos + 1], 4).reshape(b, -1, model.hkv, hd)
            a_in = o.attn_decode_qcache(q4.reshape(b, model.hq, hd), kd, vd, [pos + 1] * b).reshape(b, -1)
        ov = _lin(L["o"], a_in, out_fp32=True)
        xn, x = o.rms_norm(ov, _np(L["norm2"]), model.eps, residual_in=x)
        gf, uf = _lin(L["gate"], xn).astype(np.float32), _lin(L["up"], xn).astype(np.float32)
        a = (gf / (1 + np.exp(-gf)) * uf).astype(np.float16)
        pending = _lin(L["down"], a, out_fp32=True)
    xn, x = o.rms_norm(pending, _np(model.final_norm), model.eps, residual_in=x)
    return _lin(model.lm_head, xn).astype(np.float32)


def compute(cfg) -> dict:
    from exllamav3_amd.llama_path import SyntheticEXL3Llama
    name, K, cb, bsz = cfg[:4]
    ctx = cfg[4] if len(cfg) > 4 else None
    model = SyntheticEXL3Llama.pin_model(name, K, cb, "cpu", bsz, ctx)
    lg = oracle_logits(model)
    return {"key": SyntheticEXL3Llama.pin_key(name, K, cb, bsz, ctx), "shape": [int(v) for v in lg.shape],
            "rms": float(np.sqrt((lg.astype(np.float64) ** 2).mean())), "logits": [float(np.float32(v)) for v in lg.reshape(-1)]}


def main():
    out = {"about": "oracle logits of SyntheticEXL3Llama.pin_model (one layer + 2048-column head, host-seeded); made by tests/golden/make_bench_pins.py",
           "pins": {}}
    for cfg in CONFIGS:
        r = compute(cfg)
        out["pins"][r.pop("key")] = r
        print(cfg, "rms", round(r["rms"], 4), flush=True)
    json.dump(out, open(PINS, "w"))
    print("wrote", PINS, os.path.getsize(PINS), "bytes")


if __name__ == "__main__":
    main()
