#!/usr/bin/env python3
"""One rank's compute leg of Llama-3.1-70B at 3 bpw under TP = 8 (collectives left out: tp.OneRankOfMany): persistent step vs the launch-per-op TP branch, graph replays."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext
from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama
from exllamav3_amd.tp import OneRankOfMany
dev = torch.device("cuda:0")
ext.set_gemv_variant(1)
m = SyntheticEXL3Llama(SHAPES["llama-3.1-70b"], K=3, cb=2, device=dev, backend=OneRankOfMany(8, dev), kv_bits=4)
m.alloc_state(1)
def timed(fn, n=100):
    fn(); torch.cuda.synchronize()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn(); st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st): fn()
    torch.cuda.synchronize()
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3 / n
for rep in range(2):
    a = timed(m.decode_step_persistent); plan = m._pstep.describe() if m._pstep else None; err = m._pstep.error() if m._pstep else None
    b = timed(m.decode_step_fx)
    print(f"70B TP8 rank compute leg: persistent {a:.4f} ms (time-out {err}), launch-per-op {b:.4f} ms", flush=True)
print(plan)
