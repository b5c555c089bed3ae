#!/usr/bin/env python3
"""Ablation of the gen-2 GEMV on the lm_head shape: which pipe bounds it?  Uses diagnostics builds of the library
(build/libexl3_hip_<tag>.so, made by __graft_entry__.build_ablation; results of ablated runs are garbage by design).
Run as:  EXL3_HIP_LIB=build/libexl3_hip_<tag>.so python tools/ablate_gemv.py <tag>"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext
tag = sys.argv[1] if len(sys.argv) > 1 else "full"
dev = torch.device("cuda:0"); ext.init(0)
k, n, K = 4096, 128256, 4
g = torch.Generator(device=dev); g.manual_seed(0)
trs = [torch.randint(-32768, 32768, (k // 16, n // 16, 16 * K), dtype=torch.int16, device=dev, generator=g) for _ in range(3)]
suh = torch.ones(k, device=dev).half(); svh = torch.ones(n, device=dev).half()
x = torch.randn((1, k), device=dev).half(); y = torch.empty((1, n), dtype=torch.half, device=dev)
for cb in (2, 0):
    def body(c):
        for i in range(c): ext.exl3_gemm(x, trs[i % 3], y, suh, None, svh, -1, False, cb == 2, 0)
    body(3); torch.cuda.synchronize()
    st = torch.cuda.Stream(); gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st): body(12)
    gr.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 12
    print(json.dumps({"build": tag, "cb": cb, "us": round(us, 2), "TBps_equiv": round(k * n * K / 8 / us / 1e6, 2)}), flush=True)
