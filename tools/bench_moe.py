#!/usr/bin/env python3
"""Mixtral-8x7B-shaped sparse-MoE MLP block (hidden 4096, inter 14336, 8 experts, top-2, 4 bpw), decode bs = 1: hipGraph replay time of
router + indexed gate/up mgemm + silu_mul + weighted down mgemm (exllamav3_amd/moe_path.py)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext
from exllamav3_amd.moe_path import SyntheticEXL3MoE
dev = torch.device("cuda:0"); ext.init(0)
cb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
moes = [SyntheticEXL3MoE(4096, 14336, 8, 2, K=4, cb=cb, device=dev, seed=s) for s in range(2)]      # 2 layers' worth: cold weights
x = torch.randn((1, 4096), device=dev).half()
for m in moes: m.forward(x)
torch.cuda.synchronize()
st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    for m in moes: m.forward(x)
    st.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for rep in range(4):
            for m in moes: m.forward(x)
    g.replay(); st.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(5): g.replay()
    e1.record(st); st.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (5 * 4 * len(moes))
b = moes[0].packed_bytes_per_token()
print(json.dumps({"moe_block_us": round(us, 2), "packed_bytes": b, "TBps": round(b / us / 1e6, 3), "codebook": cb,
                  "launches": "routing + mgemm(gate,up) + silu_mul + mgemm(down)+reduce+slot-reduce"}))
