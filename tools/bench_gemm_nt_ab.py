"""Own GEMM (exl3_gemm_nt2.hip, auto form) against hipBLASLt with the measurement order varied: blocks of N launches each, alternating A B A B / B A B A, per-block times --
whether the second of a pair inherits the first one's power state.  python tools/bench_gemm_nt_ab.py [N=50]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext
dev = torch.device("cuda:0"); ext.init(0)
ext._GEMM_NT_OWN = False                              # ext.hgemm_nt = the library route here; the own kernel is called as ext.gemm_nt_mfma
torch.manual_seed(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
M = int(os.environ.get("M", "4096"))                   # rows of the activation operand (the chunk length)
def block(fn, n):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, k, n in [("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096), ("qkv", 4096, 6144)][:int(os.environ.get("NSHAPES", "4"))]:
    a = torch.randn((M, k), device=dev).half(); bt = (torch.randn((n, k), device=dev) * 0.02).half(); c = torch.empty((M, n), dtype=torch.half, device=dev)
    EPI = int(os.environ.get("EPI", "0"))                                       # 1: the residual-add form of both (c += a @ bt^T; values grow, timing only)
    if EPI == 1: c.zero_()
    ROT = int(os.environ.get("ROTATE", "1"))                                    # > 1: that many copies of a / bt walked round-robin (operands from HBM, not the 256 MB cache)
    As = [a] + [a.clone() for _ in range(ROT - 1)]; Bs = [bt] + [bt.clone() for _ in range(ROT - 1)]
    ctr = [0]
    y1 = torch.empty((M, n // 2), dtype=torch.half, device=dev) if EPI == 2 else None; y2 = torch.empty_like(y1) if EPI == 2 else None
    def own():
        i = ctr[0] % ROT; ctr[0] += 1
        ext.gemm_nt_mfma(As[i], Bs[i], y1 if EPI == 2 else c, EPI, int(os.environ.get("OWN_GEN", "0")))
    def lib():
        i = ctr[0] % ROT; ctr[0] += 1
        if EPI == 2:                                                             # (the fused silu(gate) * up epilogue against GEMM + the separate pass; timing only: rows not re-stacked)
            ext.hgemm_nt(As[i], Bs[i], c); ext.silu_mul_2d(c[:, :n // 2], c[:, n // 2:], y2)
        else:
            ext.hgemm_nt(As[i], Bs[i], c, accumulate=bool(EPI))                                      # (run with EXL3_HIP_GEMM_NT=0: hgemm_nt = the library route)
    own(); lib(); lib(); torch.cuda.synchronize()
    seq = []
    for order in ("OLOLOL", "LOLOLO"):
        for ch in order:
            seq.append((ch, round(block(own if ch == "O" else lib, N), 1)))
    o = [t for ch, t in seq if ch == "O"]; l = [t for ch, t in seq if ch == "L"]
    if os.environ.get("COMPACT"):
        print(name, "own", sorted(o)[len(o) // 2], "lib", sorted(l)[len(l) // 2], flush=True); del a, bt, c, As, Bs; continue
    print(json.dumps({"shape": name, "own_us": o, "lib_us": l, "own_med": sorted(o)[len(o) // 2], "lib_med": sorted(l)[len(l) // 2]}), flush=True)
    del a, bt, c, As, Bs
