"""hipBLASLt / rocBLAS through torch: NN (W [k][n]) vs NT (W^T [n][k]) for the prefill GEMM shapes, M = 4096, fp16."""
import torch
dev = torch.device("cuda:0")
M = 4096
def t(fn, flops):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    return ms * 1e3, flops / ms / 1e9
for (k, n) in [(4096, 4096), (4096, 1024), (4096, 14336), (14336, 4096), (4096, 6144), (4096, 28672)]:
    x = torch.randn((M, k), device=dev).half()
    w = torch.randn((k, n), device=dev).half() * 0.02
    wt = w.t().contiguous()
    y = torch.empty((M, n), device=dev, dtype=torch.half)
    fl = 2.0 * M * k * n
    a = t(lambda: torch.matmul(x, w, out=y), fl)
    b = t(lambda: torch.matmul(x, wt.t(), out=y), fl)
    print(f"k={k:6d} n={n:6d}  NN {a[0]:7.1f} us {a[1]:7.1f} TFLOP/s   NT {b[0]:7.1f} us {b[1]:7.1f} TFLOP/s")
