#!/usr/bin/env python3
"""Launch one EXL3 GEMV shape a few times (for rocprofv3 --kernel-trace / --pmc runs)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext
ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=4096); ap.add_argument("--n", type=int, default=128256)
ap.add_argument("--K", type=int, default=4); ap.add_argument("--cb", type=int, default=2); ap.add_argument("--m", type=int, default=1)
ap.add_argument("--iters", type=int, default=10); ap.add_argument("--variant", type=int, default=1); ap.add_argument("--gen", type=int, default=2)
ap.add_argument("--split", type=int, default=0); ap.add_argument("--max-waves", type=int, default=0); ap.add_argument("--copies", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0"); ext.init(0); ext.set_gemv_variant(a.variant); ext.set_gemv_max_waves(a.max_waves)
g = torch.Generator(device=dev); g.manual_seed(0)
trs = [torch.randint(-32768, 32768, (a.k // 16, a.n // 16, 16 * a.K), dtype=torch.int16, device=dev, generator=g) for _ in range(a.copies)]
suh = torch.where(torch.rand(a.k, device=dev) < 0.5, -1.0, 1.0).half(); svh = torch.where(torch.rand(a.n, device=dev) < 0.5, -1.0, 1.0).half()
x = torch.randn((a.m, a.k), device=dev).half(); y = torch.empty((a.m, a.n), dtype=torch.half, device=dev)
for i in range(a.iters):
    ext.exl3_gemm(x, trs[i % a.copies], y, suh, None, svh, -1, a.cb == 1, a.cb == 2, 0, force_split=a.split)
torch.cuda.synchronize()
print("done")
