#!/usr/bin/env python3
"""o_proj / down_proj + glue_resid as two launches vs exl3_gemv_resid (tail epilogue, XCD-local or agent-scope hand-off): GPU time per
boundary from hipGraph replays on rotating weights, Llama-3.1-8B shapes, bs 1."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext
dev = torch.device("cuda:0"); ext.init(0)
K, m = 4, int(os.environ.get("BSZ", "1"))
g = torch.Generator(device=dev); g.manual_seed(0)
def graph_us(body, n=9):
    body(3); torch.cuda.synchronize()
    st = torch.cuda.Stream(); gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(gr, stream=st): body(n)
    gr.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 5 / n
for (k, n, rot) in ((4096, 4096, False), (14336, 4096, True)):
    trs = [torch.randint(-32768, 32768, (k // 16, n // 16, 16 * K), dtype=torch.int16, device=dev, generator=g) for _ in range(3)]
    suh = torch.ones(k, device=dev).half(); svh = torch.ones(n, device=dev).half()
    x = torch.randn((m, k), device=dev).half(); xh = torch.empty_like(x); ext.had_r_128(x, xh, suh, None, 1.0)
    r = torch.zeros((m, n), device=dev).half(); ss = torch.zeros((m, n // 128), dtype=torch.float32, device=dev)
    def two(c):
        for i in range(c):
            if rot: slabs, S = ext.exl3_gemv_ex(None, [xh], None, [trs[i % 3]], None, None, None, m, False, True, ext.GEMV_IN_ROTATED | ext.GEMV_OUT_DEFERRED)
            else: slabs, S = ext.exl3_gemv_ex(x, None, None, [trs[i % 3]], None, [suh], None, m, False, True, ext.GEMV_OUT_DEFERRED)
            ext.glue_resid(slabs[0], S, svh, None, r, ss, m)
    def one(c):
        for i in range(c): ext.exl3_gemv_resid(None if rot else x, xh if rot else None, None, trs[i % 3], None if rot else suh, svh, None, m, False, True, r, ss)
    row = {"k": k, "n": n, "m": m, "gemv_plus_glue_resid_us": round(graph_us(two), 2)}
    for local in (True, False):
        ext.set_tail_xcd_local(local)
        row["gemv_resid_" + ("xcd_local" if local else "agent_scope") + "_us"] = round(graph_us(one), 2)
    ext.set_tail_xcd_local(False)
    print(json.dumps(row), flush=True)
