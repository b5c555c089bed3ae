#!/usr/bin/env python3
"""Prefill-size elementwise ops (4096 tokens of Llama-3.1-8B): GPU time and effective HBM rate."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext
dev = torch.device("cuda:0"); ext.init(0)
T, H, I = 4096, 4096, 14336
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
x = torch.randn((T, H), device=dev).half(); w = torch.ones(H, device=dev).half(); y = torch.empty_like(x); r = torch.randn((T, H), device=dev).half()
us = timed(lambda: ext.rms_norm(x, w, y, 1e-5)); print(json.dumps({"op": "rms_norm 4096x4096", "us": round(us, 1), "TBps": round(2 * T * H * 2 / us / 1e6, 2)}))
us = timed(lambda: ext.rms_norm_res_in(x, w, y, r, 1e-5)); print(json.dumps({"op": "rms_norm_res_in 4096x4096", "us": round(us, 1), "TBps": round(4 * T * H * 2 / us / 1e6, 2)}))
gu = torch.randn((T, 2 * I), device=dev).half(); a = torch.empty((T, I), dtype=torch.half, device=dev)
us = timed(lambda: ext.silu_mul_2d(gu[:, :I], gu[:, I:], a)); print(json.dumps({"op": "silu_mul_2d 4096x14336", "us": round(us, 1), "TBps": round(3 * T * I * 2 / us / 1e6, 2)}))
q = torch.randn((1, T, 32, 128), device=dev).half(); k = torch.randn((1, T, 8, 128), device=dev).half()
inv = (1.0 / (500000.0 ** (torch.arange(0, 128, 2, dtype=torch.float32) / 128))).to(dev)
us = timed(lambda: ext.rope(q, q, k, k, inv, 0, None, None, 2, 1.0)); print(json.dumps({"op": "rope 4096 tokens q32+k8 x128", "us": round(us, 1), "TBps": round(2 * T * 40 * 128 * 2 / us / 1e6, 2)}))
