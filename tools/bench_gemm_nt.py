"""Hand-written NT MFMA GEMM (GEN=2: exl3_gemm_nt2.hip, the default here; GEN=1: exl3_gemm_nt.hip) against the hipBLASLt route on the prefill shapes of Llama-3.1-8B (4096 tokens), random data:
correctness on sampled entries first, then TFLOP/s of both (HIP events, 20 launches after warm-up).  gpurun -- 'python tools/bench_gemm_nt.py'"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext
dev = torch.device("cuda:0"); ext.init(0)
ext._GEMM_NT_OWN = False                              # ext.hgemm_nt = the library route here; the own kernel is called as ext.gemm_nt_mfma
torch.manual_seed(0)

def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

out = []
M = int(os.environ.get("M", "4096"))
GEN = int(os.environ.get("GEN", "0"))
_mfma = ext.gemm_nt_mfma
ext.gemm_nt_mfma = lambda a, bt, c, epi: _mfma(a, bt, c, epi, GEN)
for name, k, n in [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096)]:
    a = torch.randn((M, k), device=dev).half()
    bt = (torch.randn((n, k), device=dev) * 0.02).half()
    c1 = torch.empty((M, n), dtype=torch.half, device=dev); c2 = torch.empty_like(c1)
    ext.gemm_nt_mfma(a, bt, c1, 0); ext.hgemm_nt(a, bt, c2); torch.cuda.synchronize()
    rows = torch.tensor([0, 1, 255, 256, 1000, M - 1], device=dev); cols = torch.tensor([0, 15, 16, 63, 64, 255, 256, 777, n - 1], device=dev)
    ref = (a[rows].float() @ bt[cols].float().T)
    e1 = float((c1[rows][:, cols].float() - ref).abs().max() / ref.abs().max()); e2 = float((c2[rows][:, cols].float() - ref).abs().max() / ref.abs().max())
    full = float((c1.float() - c2.float()).abs().max() / c2.float().abs().max())
    t1 = timeit(lambda: ext.gemm_nt_mfma(a, bt, c1, 0)); t2 = timeit(lambda: ext.hgemm_nt(a, bt, c2))
    fl = 2.0 * M * k * n
    row = {"shape": name, "m": M, "k": k, "n": n, "err_mine": round(e1, 5), "err_lib": round(e2, 5), "max_diff_vs_lib": round(full, 5),
           "gen": GEN, "mine_us": round(t1, 1), "lib_us": round(t2, 1), "mine_tflops": round(fl / t1 / 1e6, 1), "lib_tflops": round(fl / t2 / 1e6, 1)}
    if name == "gate_up":
        # fused silu * mul epilogue against GEMM + silu_mul_2d: bt rows re-stacked per 256-row tile as 128 gate | 128 up
        inter = n // 2
        bt_s = bt.view(2, inter // 128, 128, k).permute(1, 0, 2, 3).reshape(n, k).contiguous()
        y1 = torch.empty((M, inter), dtype=torch.half, device=dev)
        ext.gemm_nt_mfma(a, bt_s, y1, 2); torch.cuda.synchronize()
        g = c2[:, :inter].float(); u = c2[:, inter:].float()
        yr = (g / (1 + torch.exp(-g)) * u)
        row["silu_err"] = round(float((y1.float() - yr).abs().max() / yr.abs().max()), 5)
        row["mine_silu_us"] = round(timeit(lambda: ext.gemm_nt_mfma(a, bt_s, y1, 2)), 1)
        y2 = torch.empty_like(y1)
        def lib_silu():
            ext.hgemm_nt(a, bt, c2); ext.silu_mul_2d(c2[:, :inter], c2[:, inter:], y2)
        try:
            row["lib_plus_silu_us"] = round(timeit(lib_silu), 1)
        except Exception as e:
            row["lib_plus_silu_us"] = repr(e)[:80]
    if name in ("o", "down"):
        r1 = torch.randn((M, n), device=dev).half(); r2 = r1.clone()
        ext.gemm_nt_mfma(a, bt, r1, 1); ext.hgemm_nt(a, bt, r2, accumulate=True); torch.cuda.synchronize()
        row["acc_diff_vs_lib"] = round(float((r1.float() - r2.float()).abs().max() / r2.float().abs().max()), 5)
    print(json.dumps(row), flush=True)
    out.append(row)
    del a, bt, c1, c2
os.makedirs("gpurun_out/r6", exist_ok=True)
json.dump(out, open(f"gpurun_out/r6/bench_gemm_nt_gen{GEN}.json", "w"), indent=1)
