#!/usr/bin/env python3
"""LinearEXL3 forward at 16..512 rows: small-m kernel (generation 3 passes) vs the reconstruct + hgemm routes (plain < 1024 rows route and the
fused reconstruct_had + NT GEMM route forced on), Llama-3.1-8B shapes.  Decides where linear.py should switch."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext, linear
from exllamav3_amd.linear import LinearEXL3
dev = torch.device("cuda:0"); ext.init(0)
K = 4
g = torch.Generator(device=dev); g.manual_seed(0)

def timed(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

for (k, n) in ((4096, 6144), (4096, 14336), (14336, 4096), (4096, 4096)):
    tr = torch.randint(-32768, 32768, (k // 16, n // 16, 16 * K), dtype=torch.int16, device=dev, generator=g)
    lin = LinearEXL3(k, n, tr, torch.ones(k, device=dev).half(), torch.ones(n, device=dev).half(), mul1=True)
    for rows in [int(v) for v in sys.argv[1:]] or [16, 32, 64, 96, 128, 144, 192, 256, 512]:
        x = torch.randn((rows, k), device=dev).half()
        row = {"k": k, "n": n, "rows": rows}
        row["small_m_us"] = round(timed(lambda: lin.forward(x, {"no_reconstruct": True})), 1)
        row["reconstruct_us"] = round(timed(lambda: lin.forward(x, {"reconstruct": True})), 1)
        saved = linear.FUSED_RECONSTRUCT_MIN_ROWS
        linear.FUSED_RECONSTRUCT_MIN_ROWS = 1
        row["fused_reconstruct_us"] = round(timed(lambda: lin.forward(x, {"reconstruct": True})), 1)
        linear.FUSED_RECONSTRUCT_MIN_ROWS = saved
        print(json.dumps(row), flush=True)
