"""Times exl3_attn_prefill_paged alone: causal attention of a T-token chunk (Llama-3.1-8B heads by default) over paged fp16 K/V.
   python tools/bench_attn_prefill.py [T] [ctx]   -> µs per launch and TFLOP/s on the causal 4 * hd * hq * (ctx * T + T^2 / 2) flops"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exllamav3_amd import ext

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hq, hkv, hd, page = int(os.environ.get("HQ", 32)), int(os.environ.get("HKV", 8)), int(os.environ.get("HD", 128)), 256      # HQ / HKV / HD: other head counts / head_dim (e.g. 28 / 4, 64)
dev = torch.device("cuda:0")
L = ctx + T
pps = (L + page - 1) // page
q = torch.randn((1, T, hq, hd), dtype=torch.half, device=dev)
k = torch.randn((pps, page, hkv, hd), dtype=torch.half, device=dev)
v = torch.randn_like(k)
out = torch.empty_like(q)
bt = torch.arange(pps, dtype=torch.int32, device=dev).view(1, pps)
lens = torch.tensor([L], dtype=torch.int32, device=dev)
for _ in range(3):
    ext.attn_prefill_paged(q, out, k, v, bt, lens)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    ext.attn_prefill_paged(q, out, k, v, bt, lens)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
flops = 4.0 * hd * hq * (ctx * T + T * (T + 1) / 2)
print(f"attn_prefill hq={hq} hkv={hkv} hd={hd} T={T} ctx={ctx}: {us:.1f} us, {flops / us * 1e-6:.1f} TFLOP/s (causal flops)")
