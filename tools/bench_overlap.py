#!/usr/bin/env python3
"""decode_step_overlap (glue kernels on a second stream, consumer GEMVs waiting on a flag) vs decode_step_fused: bit identity of the logits,
the give-up word, and step time eager / under hipGraph replay.   LAYERS=32 BSZ=1 python tools/bench_overlap.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext
from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama

dev = torch.device("cuda:0")
layers = int(os.environ.get("LAYERS", "32")); bsz = int(os.environ.get("BSZ", "1"))
model = SyntheticEXL3Llama(SHAPES[os.environ.get("MODEL", "llama-3.1-8b")], K=4, cb=2, device=dev, kv_bits=4, layers=layers)
model.alloc_state(bsz)
model.with_attention = os.environ.get("ATTN", "0") == "1"
ref = model.decode_step_fused().clone(); torch.cuda.synchronize()
out = {"layers": layers, "bsz": bsz}
got = model.decode_step_overlap().clone(); torch.cuda.synchronize()
out["eager_identical"] = bool(torch.equal(ref, got)); out["eager_err"] = int(model.ovl_err.item())

def graph_of(fn):
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn(); st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            fn()
    torch.cuda.synchronize()
    return g

def time_graph(g, n=30):
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

gf = graph_of(model.decode_step_fused)
out["fused_ms"] = round(time_graph(gf), 4)
go = graph_of(model.decode_step_overlap)
model.logits.zero_(); go.replay(); torch.cuda.synchronize()
out["graph_identical"] = bool(torch.equal(ref, model.logits)); out["graph_err"] = int(model.ovl_err.item())
out["overlap_ms"] = round(time_graph(go), 4)
out["graph_err_after_timing"] = int(model.ovl_err.item())
model.logits.zero_(); go.replay(); torch.cuda.synchronize()
out["graph_identical_after_timing"] = bool(torch.equal(ref, model.logits))
out["fused_ms_again"] = round(time_graph(gf), 4)
out["ratio"] = round(out["overlap_ms"] / out["fused_ms"], 4)
print(json.dumps(out))
