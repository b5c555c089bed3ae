#!/usr/bin/env python3
"""One tensor-parallel rank's compute leg on one GPU (tp.OneRankOfMany: the exchange left out of the all-reduce launches): glue TP step vs the fx TP
step (round 4), Llama-3.1-70B at 3 bpw TP = 8 and Llama-3.1-8B TP = 2 / 8.  python tools/r4_tp_rank_time.py"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext
from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama
from exllamav3_amd.tp import OneRankOfMany
dev = torch.device("cuda:0"); ext.init(0)

def timed(fn):
    fn(); torch.cuda.synchronize()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn(); st.synchronize(); g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st): fn()
    torch.cuda.synchronize()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(30): g.replay()
    torch.cuda.synchronize(); return round((time.perf_counter() - t) / 30 * 1e3, 4)

for name, K, tp in (("llama-3.1-70b", 3, 8), ("llama-3.1-8b", 4, 2), ("llama-3.1-8b", 4, 8)):
    m = SyntheticEXL3Llama(SHAPES[name], K=K, cb=2, device=dev, backend=OneRankOfMany(tp, dev), kv_bits=4)
    m.alloc_state(1)
    row = {"model": name, "tp": tp, "glue_ms": timed(m.decode_step_fused)}
    for act in (False, True):
        m.fx_tp_act_in_gemv = act
        row["fx_act_in_gemv_%d_ms" % act] = timed(m.decode_step_fx)
    m.backend.ipc_like = False
    m.fx_tp_act_in_gemv = False
    row["fx_dense_partials_ms"] = timed(m.decode_step_fx)
    print(row, flush=True)
    del m; torch.cuda.empty_cache()
