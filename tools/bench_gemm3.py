#!/usr/bin/env python3
"""Generation 2 vs 3 small-m GEMM on the Llama-3.1-8B decode shapes (hipGraph, rotating weight copies).  usage: bench_gemm3.py [m ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext
dev = torch.device("cuda:0"); ext.init(0)
K = 4
g = torch.Generator(device=dev); g.manual_seed(0)
ms = [int(v) for v in sys.argv[1:]] or [5, 8, 12, 16, 32]
for (k, n) in ((4096, 6144), (4096, 28672), (14336, 4096), (4096, 4096)):
    trs = [torch.randint(-32768, 32768, (k // 16, n // 16, 16 * K), dtype=torch.int16, device=dev, generator=g) for _ in range(3)]
    suh = torch.ones(k, device=dev).half(); svh = torch.ones(n, device=dev).half()
    for m in ms:
        x = torch.randn((m, k), device=dev).half(); y = torch.empty((m, n), dtype=torch.half, device=dev)
        row = {"k": k, "n": n, "m": m}
        for name, mr in (("gen2", 0), ("gen3", 5)):
            if name == "gen2" and m > 16: continue
            ext.set_gemm3_min_rows(mr)
            def body(c):
                for i in range(c): ext.exl3_gemm(x, trs[i % 3], y, suh, None, svh, -1, False, True, 0)
            body(3); torch.cuda.synchronize()
            st = torch.cuda.Stream(); gr = torch.cuda.CUDAGraph()
            with torch.cuda.stream(st):
                with torch.cuda.graph(gr, stream=st): body(9)
            gr.replay(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            row[name + "_us"] = round(e0.elapsed_time(e1) * 1e3 / 9, 2)
        print(json.dumps(row), flush=True)
    del trs
