"""Run by tests/test_oracle_pins.py in a subprocess whose PYTHONPATH = [stub dir, repo, /root/reference, /root/reference/tests]: executes the REFERENCE's
own rope tests (tests/test_rope.py: test_rope over its 13 shapes x 2 styles x head norm x 4 position modes, test_rope_multidim, test_rope_llama4_scaling,
and whatever follows in that file) on the CPU with `ext.rope` replaced by an adapter onto oracle.exl3_oracle.rope -- i.e. the oracle's restatement of
rope.cu is checked by the reference's own torch restatements and tolerances (3e-3; keys bitwise unscaled under the llama-4 scale).  No GPU, no native module."""
import sys, inspect, itertools
import numpy as np
import torch
from oracle import exl3_oracle as o
import exllamav3_ext                      # the stub package (exllamav3_amd/stub): the module object the reference's util/rope.py binds as `ext`
from exllamav3.util.rope import RopeStyle

calls = {"n": 0}


def rope_adapter(q, out_q, k, out_k, inv_freq, position, positions, position_ids, rope_mode, attn_factor, q_norm, k_norm, norm_eps, norm_constant_bias,
                 l4_beta, l4_orig, post_rope_norm, rotate_dims, rotate_offset):
    calls["n"] += 1
    npy = lambda t: None if t is None else t.detach().cpu().numpy()
    bf16 = q_norm is not None and q_norm.dtype == torch.bfloat16
    nw = lambda t: None if t is None else (t.float().numpy() if bf16 else t.numpy())
    rq, rk = o.rope(npy(q), npy(k), inv_freq.float().numpy(), position=int(position), positions=npy(positions), position_ids=npy(position_ids),
                    rope_mode=int(rope_mode), attn_factor=float(attn_factor), q_norm=nw(q_norm), k_norm=nw(k_norm), norm_eps=float(norm_eps),
                    norm_constant_bias=float(norm_constant_bias), l4_beta=float(l4_beta), l4_orig=int(l4_orig), post_rope_norm=bool(post_rope_norm),
                    rotate_dims=int(rotate_dims), rotate_offset=int(rotate_offset), norm_bf16=bf16)
    out_q.copy_(torch.from_numpy(rq))
    if k is not None:
        out_k.copy_(torch.from_numpy(rk))


exllamav3_ext.rope = rope_adapter
import exllamav3.util.rope as ref_rope_mod
ref_rope_mod.ext.rope = rope_adapter
import test_rope as T                     # /root/reference/tests/test_rope.py
T.device = "cpu"
T.ext.rope = rope_adapter

ran = []
for name, fn in inspect.getmembers(T, inspect.isfunction):
    if not name.startswith("test_"):
        continue
    marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
    names = [m.args[0] for m in marks]
    values = [list(m.args[1]) for m in marks]
    inner = fn
    while hasattr(inner, "__wrapped__"):
        inner = inner.__wrapped__
    n0 = calls["n"]
    for combo in itertools.product(*values):
        kw = dict(zip(names, combo))
        if name == "test_rope" and kw["qk_dim"][0][1] * kw["qk_dim"][0][2] > 20000:
            continue                       # (1, 864, 128, 128) and (1, 600, 80, 256): minutes of numpy, no new code path
        with torch.inference_mode():
            inner(**kw)
    ran.append((name, calls["n"] - n0))
assert all(c > 0 for _, c in ran), ran
print("REFERENCE_ROPE_TESTS_OK", ran)
