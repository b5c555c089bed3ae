import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from exllamav3_amd import ext
dev = torch.device("cuda:0"); ext.init(0)
def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (M, n, k) in [(256, 256, 4096), (512, 512, 4096), (2048, 2048, 4096), (4096, 4096, 4096), (4096, 4096, 8192)]:
    a = torch.randn((M, k), device=dev).half(); bt = (torch.randn((n, k), device=dev) * 0.02).half(); c = torch.empty((M, n), dtype=torch.half, device=dev)
    import ctypes
    from exllamav3_amd import _lib
    for mode in (0,):
        f = lambda: _lib.lib().exl3_gemm_nt_mfma(a.data_ptr(), a.stride(0), bt.data_ptr(), bt.stride(0), c.data_ptr(), c.stride(0), M, k, n, mode << 8, None)
        t = timeit(f)
        print(M, n, k, "mode", mode, "us", round(t, 1), "per K-tile ns", round(t * 1e3 / (k // 64), 1), "TF", round(2.0 * M * n * k / t / 1e6, 1))
