"""Timing of the generation-2 GEMM's diagnostic K-loop variants (tools/r6_gn2_variants.sh builds build/alt_gn2var/libexl3_hip.so):
python tools/bench_gn2_variants.py <n variants> -> one line per variant and shape (us, TFLOP/s as if the full work were done).  Variant 0 = the product loop."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch
    from exllamav3_amd import ext
    dev = torch.device("cuda:0"); ext.init(0); torch.manual_seed(0)
    res = {}
    for name, k, n in [("o", 4096, 4096), ("down", 14336, 4096), ("gate_up", 4096, 28672), ("qkv", 4096, 6144)]:
        a = torch.randn((4096, k), device=dev).half(); bt = (torch.randn((n, k), device=dev) * 0.02).half(); c = torch.empty((4096, n), dtype=torch.half, device=dev)
        f = lambda: ext.gemm_nt_mfma(a, bt, c, 0, int(os.environ.get('GEN', '2')))
        f(); f(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        res[name] = [round(us, 1), round(2.0 * 4096 * k * n / us / 1e6, 1)]
        try:
            import ctypes, statistics
            L = ctypes.CDLL(os.environ["EXL3_HIP_LIB"]); nwg = 16 * (n // 256)
            buf = (ctypes.c_ulonglong * (2 * nwg))()
            if nwg <= 4096 and L.exl3_gn2_dbg_read(buf, nwg) == 0:
                tk = [buf[2 * i] for i in range(nwg)]; rt = [buf[2 * i + 1] for i in range(nwg)]
                res[name] += [int(statistics.median(tk)), round(statistics.median(t / max(r, 1) * 100 for t, r in zip(tk, rt)))]      # ticks to the end of the K-loop, MHz
        except Exception as e:
            res[name].append(repr(e)[:60])
        del a, bt, c
    print(json.dumps({"variant": int(sys.argv[2]), **res}), flush=True)
else:
    nv = int(sys.argv[1]); rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    lib = os.environ.get("GN2_LIB", os.path.join(ROOT, "build", "alt_gn2var", "libexl3_hip.so"))
    for r in range(rounds):
        for v in range(nv + 1):
            env = dict(os.environ, EXL3_HIP_LIB=lib, EXL3_GN2_VARIANT=str(v))
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", str(v)], env=env, timeout=120)
