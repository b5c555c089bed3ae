"""o_proj's split-k count with the q|k|v epilogue inside its launch (exl3_gemv_ex_qkvm): more slices = fewer preparation tasks per workgroup but more
workgroups finishing the same q block.  Llama-3.1-8B bs 1 fx step, hipGraph replay, one process; prints ms per step for each forced split."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama

dev = torch.device("cuda:0")
m = SyntheticEXL3Llama(SHAPES["llama-3.1-8b"], K=4, cb=2, device=dev, kv_bits=4)
m.alloc_state(1)


def timed(steps=60):
    m.decode_step_fx(); torch.cuda.synchronize()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        m.decode_step_fx(); st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            m.decode_step_fx()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps


res = {}
for rep in range(2):
    for so in (0, 4, 8, 16, 32):
        type(m).split = dict(type(m).split, o=so)
        res.setdefault(f"qkvm_o{so}", []).append(round(timed(), 4))
    m.qkv_in_oproj = False
    type(m).split = dict(type(m).split, o=0)
    res.setdefault("glue_qkv_o_default", []).append(round(timed(), 4))
    m.qkv_in_oproj = True
print(json.dumps(res))
