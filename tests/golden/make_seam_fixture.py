"""Generates tests/golden/ref_seam_calls.npz (run on a box that has /root/reference; the GPU box does not):

    PYTHONPATH=exllamav3_amd/stub:.:/root/reference python tests/golden/make_seam_fixture.py

The REFERENCE's own Python classes -- modules.quant.exl3.LinearEXL3 (kernel route, reconstruct + hgemm, fused reconstruct), modules.rmsnorm.RMSNorm,
util.rope.RoPE -- are driven on CPU tensors over this build's `exllamav3_ext` module with a RECORDING op layer: every leaf op they reach
(exl3_gemm, had_r_128, reconstruct*, hgemm, rms_norm, rope) is logged with its exact argument list -- tensors as (storage, offset, shape, strides,
dtype), so views and in-place reuse alias exactly as in the reference's code -- instead of executed.  Stored with each case: the initial contents of
every storage, the call sequence, which tensor the reference method returned, and the ORACLE's result for the same inputs.
tests/test_gpu_reference_seam.py replays the sequences on the GPU (no reference tree needed there) and compares with the stored oracle result:
the reference's call sequence, argument order and aliasing, executed by the HIP kernels, produce the oracle's values."""
import io
import json
import os
import sys
import types
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import exl3_oracle as o
import exllamav3_ext                                                    # the stub: this build's module under the reference's name
import exllamav3_amd.ext as impl
assert exllamav3_ext.__implementation__ is impl

# leaf ops the hot-path classes reach, and which positional arguments they WRITE (their first appearance needs no stored contents)
LEAF = {"exl3_gemm": (2, 4), "had_r_128": (1,), "reconstruct": (0,), "reconstruct_slice": (0,), "reconstruct_had_slice": (0,), "hgemm": (2,),
        "rms_norm": (2,), "rope": (1, 3)}
DT = {torch.float16: "f2", torch.float32: "f4", torch.int16: "i2", torch.int32: "i4", torch.int64: "i8", torch.bfloat16: "bf2"}


class Recorder:
    def __init__(self):
        self.storages, self.calls, self.keep = {}, [], []

    def tdesc(self, t, written):
        st = t.untyped_storage()
        key = st.data_ptr()
        if key not in self.storages:
            self.keep.append(t)                                          # keep the storage alive: data_ptr stays unique
            raw = np.frombuffer(bytes(st), dtype=np.uint8).copy()       # contents at first appearance
            self.storages[key] = {"id": len(self.storages), "nbytes": st.nbytes(), "init": None if written else raw}
        s = self.storages[key]
        return {"s": s["id"], "off": int(t.storage_offset()), "shape": list(t.shape), "strides": list(t.stride()), "dt": DT[t.dtype]}

    def arg(self, a, written):
        if isinstance(a, torch.Tensor): return {"t": self.tdesc(a, written)}
        if a is None or isinstance(a, (bool, int, float, str)): return {"v": a}
        raise TypeError(f"unrecordable argument {type(a)}")

    def wrap(self, name):
        out_idx = LEAF[name]
        def fn(*args, **kwargs):
            rec = {"op": name, "args": [self.arg(a, i in out_idx) for i, a in enumerate(args)],
                   "kwargs": {k: self.arg(v, False) for k, v in kwargs.items()}}
            self.calls.append(rec)
            return 0
        return fn


def run_case(build):
    """build(rec) -> (returned tensor, oracle result ndarray, tolerance).  Ops are recorded, not executed."""
    rec = Recorder()
    saved = {}
    for name in LEAF:
        saved[name] = (getattr(exllamav3_ext, name), getattr(impl, name))
        w = rec.wrap(name)
        setattr(exllamav3_ext, name, w); setattr(impl, name, w)          # the stub's namespace (reference modules) and the mirror's own (BC_* runners)
    try:
        ret, expect, tol = build(rec)
    finally:
        for name, (a, b) in saved.items():
            setattr(exllamav3_ext, name, a); setattr(impl, name, b)
    return {"calls": rec.calls, "result": rec.tdesc(ret, True), "tol": tol,
            "storages": [{"id": s["id"], "nbytes": s["nbytes"], "has_init": s["init"] is not None} for s in rec.storages.values()]}, \
           {f"s{s['id']}": s["init"] for s in rec.storages.values() if s["init"] is not None}, expect


def main():
    from exllamav3.modules.quant.exl3 import LinearEXL3
    from exllamav3.modules.rmsnorm import RMSNorm
    from exllamav3.util.rope import RoPE, RopeSettings, RopeStyle
    cases, arrays = {}, {}
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))

    def add(name, build):
        meta, inits, expect = run_case(build)
        cases[name] = meta
        for k, v in inits.items(): arrays[f"{name}/{k}"] = v
        arrays[f"{name}/expect"] = expect.astype(np.float16)

    # ---- LinearEXL3.forward (modules/quant/exl3.py:112-218): kernel route (1 and 16 rows), reconstruct + hgemm (145), fused reconstruct (1030)
    k, n, K = 256, 256, 4
    tr, su, sv = o.synth_linear(k, n, K, seed=3, realistic=True)
    for cb, marker in ((2, {"mul1": torch.zeros(1, dtype=torch.int32)}), (0, {})):
        for rows in ((1, 16, 145, 1030) if cb == 2 else (1, 16, 145)):
            x = np.random.default_rng(rows).standard_normal((rows, k)).astype(np.float16)
            def build(rec, x=x, cb=cb, marker=marker, rows=rows):
                lin = LinearEXL3(None, k, n, suh=T(su), svh=T(sv), trellis=T(tr), key="seam.test", **marker)
                y = lin.forward(T(x), {})
                return y, o.linear_forward(x, tr, su, sv, K, cb).astype(np.float32), (1e-2 if rows <= 144 else 2e-2)
            add(f"linear_cb{cb}_rows{rows}", build)

    # ---- RMSNorm.forward (modules/rmsnorm.py:98-160)
    xs = (np.random.default_rng(1).standard_normal((3, 512)) * 2).astype(np.float16)
    wn = (1 + 0.1 * np.random.default_rng(2).standard_normal(512)).astype(np.float16)
    def build_norm(rec):
        ns = types.SimpleNamespace(weight=T(wn), rms_norm_eps=1e-5, constant_bias=0.0, constant_scale=1.0, span_heads=False, out_dtype=torch.half, key="n")
        y = RMSNorm.forward(ns, T(xs), {})
        return y, o.rms_norm(xs, wn, 1e-5).astype(np.float32), 2e-3
    add("rmsnorm", build_norm)

    # ---- RoPE.apply (util/rope.py:469-530), llama3 scaling, NEOX: q
    rs = RopeSettings(head_dim=128, rope_theta=500000.0, rope_scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0,
                      "high_freq_factor": 4.0, "original_max_position_embeddings": 8192}, max_position_embeddings=131072, rope_style=RopeStyle.NEOX)
    q = np.random.default_rng(5).standard_normal((2, 5, 4, 128)).astype(np.float16)
    kk = np.random.default_rng(6).standard_normal((2, 5, 2, 128)).astype(np.float16)
    for which in (0, 1):
        def build_rope(rec, which=which):
            rope = RoPE("cpu", rs)
            qo, ko = rope.apply(T(q), T(kk), position=37)
            rq, rk = o.rope(q, kk, rope.inv_freq.float().numpy(), position=37, rope_mode=o.ROPE_NEOX)
            return (qo, ko)[which], (rq, rk)[which].astype(np.float32), 3e-3
        add(f"rope_{'qk'[which]}", build_rope)

    out = os.path.join(ROOT, "tests", "golden", "ref_seam_calls.npz")
    arrays["meta"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    np.savez_compressed(out, **arrays)
    nbytes = os.path.getsize(out)
    print(f"wrote {out}: {len(cases)} cases, {sum(len(c['calls']) for c in cases.values())} recorded calls, {nbytes / 1e6:.2f} MB")
    for name, c in cases.items():
        print(f"  {name}: " + " -> ".join(call["op"] for call in c["calls"]))


if __name__ == "__main__":
    main()
