#!/usr/bin/env python3
"""reconstruct_had_slice_t (W^T, the prefill route) on the Llama-3.1-8B linears: GPU time from events around 9 calls on rotating buffers."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext
dev = torch.device("cuda:0"); ext.init(0)
K = 4
g = torch.Generator(device=dev); g.manual_seed(0)
for (k, n) in ((4096, 1024), (4096, 4096), (4096, 14336), (14336, 4096)):
    trs = [torch.randint(-32768, 32768, (k // 16, n // 16, 16 * K), dtype=torch.int16, device=dev, generator=g) for _ in range(3)]
    suh = torch.ones(k, device=dev).half(); svh = torch.ones(n, device=dev).half()
    ws = [torch.empty((n, k), dtype=torch.half, device=dev) for _ in range(3)]
    for i in range(3): ext.reconstruct_had_slice_t(ws[i], trs[i], suh, svh, K, False, True, 0)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(9): ext.reconstruct_had_slice_t(ws[i % 3], trs[i % 3], suh, svh, K, False, True, 0)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 9
    print(json.dumps({"k": k, "n": n, "us": round(us, 1), "TBps": round((k * n * 2 + k * n * K / 8) / us / 1e6, 2)}), flush=True)
