"""Per-call-type GPU time of the decode GEMV launches (graph replays bracketed by events), for the three pipelines."""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama

dev = torch.device("cuda:0")
model = SyntheticEXL3Llama(SHAPES["llama-3.1-8b"], K=4, cb=2, device=dev, kv_bits=4, layers=int(sys.argv[1]) if len(sys.argv) > 1 else 8)
model.alloc_state(int(os.environ.get("BSZ", "1")))
import os
from exllamav3_amd import ext
if os.environ.get("MAXW"): ext.set_gemv_max_waves(int(os.environ["MAXW"]))
if os.environ.get("G3MIN"): ext.set_gemm3_min_rows(int(os.environ["G3MIN"]))
for pipe in os.environ.get("PIPES", "glue,tail").split(","):
    {"glue": model.decode_step_fused, "tail": model.decode_step_tail}[pipe]()
    torch.cuda.synchronize()
    calls = model.gemv_calls(pipe)
    names = ["qkv", "o", "gate_up", "down"]
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for i in range(4):
            grp = calls[i:len(calls) - 1:4]
            for c in grp: c()
            st.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                for c in grp: c()
            g.replay(); st.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10): g.replay()
            e1.record(st); st.synchronize()
            print(f"{pipe:5s} {names[i]:8s} {e0.elapsed_time(e1) * 1e3 / 10 / len(grp):7.2f} us/launch")
