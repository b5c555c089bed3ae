#!/usr/bin/env python3
"""Time exl3_gemm on arbitrary (k, n) shapes (hipGraph, GPU-side events, rotating weight copies)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext
dev = torch.device("cuda:0"); ext.init(0)
K = 4
g = torch.Generator(device=dev); g.manual_seed(0)
for spec in sys.argv[1:]:
    k, n, mw, split = [int(v) for v in spec.split(",")]
    ext.set_gemv_max_waves(mw)
    trs = [torch.randint(-32768, 32768, (k // 16, n // 16, 16 * K), dtype=torch.int16, device=dev, generator=g) for _ in range(3)]
    suh = torch.ones(k, device=dev).half(); svh = torch.ones(n, device=dev).half()
    x = torch.randn((1, k), device=dev).half(); y = torch.empty((1, n), dtype=torch.half, device=dev)
    def body(c):
        for i in range(c): ext.exl3_gemm(x, trs[i % 3], y, suh, None, svh, -1, False, True, 0, force_split=split)
    body(3); torch.cuda.synchronize()
    st = torch.cuda.Stream(); gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st): body(9)
    gr.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 9
    print(json.dumps({"k": k, "n": n, "max_waves": mw, "split": split, "us": round(us, 2), "TBps": round(k * n * K / 8 / us / 1e6, 2)}), flush=True)
    del trs
