#!/usr/bin/env python3
"""Per-kernel timing of the EXL3 GEMV/GEMM on the Llama-3.1-8B shapes (science/qgemm_benchmark.py:15-34 shapes),
with weight-buffer rotation through > 512 MB so neither L2 nor the 256 MiB Infinity Cache serves the trellis
(qgemm_benchmark.py:73-81), 10 warmup + 60 timed launches (:11,98-109).  Prints one JSON line per configuration.

Usage: python tools/bench_kernels.py [--quick] [--variants 0,1] [--cbs 0,2] [--ms 1,16] [--K 4] [--splits 0]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from exllamav3_amd import ext

SHAPES = [("q/o", 4096, 4096), ("k/v", 4096, 1024), ("gate/up", 4096, 14336), ("down", 14336, 4096), ("lm_head", 4096, 128256)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--variants", default="0,1")
    ap.add_argument("--cbs", default="0,2")
    ap.add_argument("--ms", default="1,16")
    ap.add_argument("--K", default="4")
    ap.add_argument("--splits", default="0")
    ap.add_argument("--shapes", default="")
    ap.add_argument("--gens", default="2")
    ap.add_argument("--eager", action="store_true", help="time eager launches (host-bound for small kernels) instead of a hipGraph")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ext.init(0)
    gen = torch.Generator(device="cpu").manual_seed(0)
    warm, iters = (3, 20) if args.quick else (10, 60)
    shapes = [s for s in SHAPES if not args.shapes or s[0] in args.shapes.split(",")]
    for K in [int(v) for v in args.K.split(",")]:
        for (name, k, n) in shapes:
            wbytes = k * n * K // 8
            copies = max(2, min(80, (600 << 20) // wbytes + 1))
            trs = [torch.randint(-32768, 32767, (k // 16, n // 16, 16 * K), dtype=torch.int16, generator=gen).to(dev) for _ in range(min(copies, 4))]
            while len(trs) < copies:
                trs.append(trs[len(trs) % 4].clone())
            suh = torch.sign(torch.randn(k, generator=gen)).half().to(dev)
            svh = torch.sign(torch.randn(n, generator=gen)).half().to(dev)
            for m in [int(v) for v in args.ms.split(",")]:
                x = torch.randn((m, k), generator=gen).half().to(dev)
                y = torch.empty((m, n), dtype=torch.half, device=dev)
                for cb in [int(v) for v in args.cbs.split(",")]:
                  for gg in [int(v) for v in args.gens.split(",")]:
                    for var in [int(v) for v in args.variants.split(",")]:
                        for split in [int(v) for v in args.splits.split(",")]:
                            ext.set_gemv_variant(var)

                            def body(count):
                                for i in range(count):
                                    ext.exl3_gemm(x, trs[i % copies], y, suh, None, svh, -1, cb == 1, cb == 2, 0, force_split=split)
                            body(warm)
                            torch.cuda.synchronize()
                            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                            if args.eager:
                                e0.record(); body(iters); e1.record()
                                torch.cuda.synchronize()
                            else:
                                st = torch.cuda.Stream()
                                g = torch.cuda.CUDAGraph()
                                with torch.cuda.stream(st):
                                    with torch.cuda.graph(g, stream=st):
                                        body(iters)
                                g.replay(); torch.cuda.synchronize()
                                e0.record(); g.replay(); e1.record()
                                torch.cuda.synchronize()
                            us = e0.elapsed_time(e1) * 1e3 / iters
                            algo = wbytes + 2 * (k + n) + 2 * m * (k + n)
                            print(json.dumps({"shape": name, "k": k, "n": n, "K": K, "m": m, "cb": cb, "gen": gg, "variant": var, "split": split,
                                              "us": round(us, 2), "GBps": round(algo / us / 1e3, 1),
                                              "frac_of_8TBps": round(algo / us / 1e3 / 8000, 3)}), flush=True)
            del trs
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
