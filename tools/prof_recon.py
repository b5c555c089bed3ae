#!/usr/bin/env python3
"""reconstruct_had_slice on the gate_proj shape (4096 x 14336, K = 4, mul1) for rocprofv3 runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext
dev = torch.device("cuda:0"); ext.init(0)
k, n, K = 4096, 14336, 4
g = torch.Generator(device=dev); g.manual_seed(0)
trs = [torch.randint(-32768, 32768, (k // 16, n // 16, 16 * K), dtype=torch.int16, device=dev, generator=g) for _ in range(3)]
suh = torch.ones(k, device=dev).half(); svh = torch.ones(n, device=dev).half()
ws = [torch.empty((k, n), dtype=torch.half, device=dev) for _ in range(3)]
for i in range(9):
    ext.reconstruct_had_slice(ws[i % 3], trs[i % 3], suh, svh, K, False, True, 0)
torch.cuda.synchronize()
print("done")
