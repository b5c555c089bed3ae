#!/usr/bin/env python3
"""bs > 4 decode: glue_resid_rotate (8 launches / layer) vs separate glue_resid + glue_rotate (10): step time under hipGraph replay.
   BSZ=16 python tools/bench_fold.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama

dev = torch.device("cuda:0")
bsz = int(os.environ.get("BSZ", "16"))
model = SyntheticEXL3Llama(SHAPES[os.environ.get("MODEL", "llama-3.1-8b")], K=4, cb=2, device=dev, kv_bits=4, layers=int(os.environ.get("LAYERS", "32")))
model.alloc_state(bsz)

def graph_of(fn):
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn(); st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            fn()
    torch.cuda.synchronize()
    return g

def time_graph(g, n=20):
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): g.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best

out = {"bsz": bsz}
for fold in (False, True, False, True):
    model.fold_rotate = fold
    g = graph_of(model.decode_step_fused)
    out.setdefault("folded_ms" if fold else "separate_ms", []).append(round(time_graph(g), 4))
print(json.dumps(out))
