"""Decode step time vs threads per workgroup of the glue kernels (exl3_set_glue_threads).  usage: bench_glue_threads.py [bsz]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exllamav3_amd import _lib
from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama
bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda:0")
model = SyntheticEXL3Llama(SHAPES["llama-3.1-8b"], K=4, cb=2, device=dev, kv_bits=4, layers=16)
model.alloc_state(bsz)
for th in (256, 128, 64, 0):
    _lib.lib().exl3_set_glue_threads(th)
    model.decode_step_fused(); torch.cuda.synchronize()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            model.decode_step_fused()
        g.replay(); st.synchronize()
        best = 1e9
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10): g.replay()
            e1.record(st); st.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 10)
    print(f"bs {bsz} glue threads {th:3d}: {best / 16:.2f} us/layer")
