"""Whole-step time (hipGraph replay, full model) for combinations of split-k factors, alternated with the heuristic to cancel drift.
   python tools/sweep_split_combo.py '[{"qkv":16},{"qkv":16,"o":16},...]'   env: MODEL, BSZ"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama

dev = torch.device("cuda:0")
model = SyntheticEXL3Llama(SHAPES[os.environ.get("MODEL", "llama-3.1-8b")], K=4, cb=2, device=dev, kv_bits=4)
model.alloc_state(int(os.environ.get("BSZ", "1")))

def step_ms():
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
            for _ in range(20): g.replay()
            e1.record(st); st.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
    return best

combos = json.loads(sys.argv[1])
base = dict(SyntheticEXL3Llama.split)
step_ms()
for c in combos:
    model.split = dict(base); b = step_ms()
    model.split = dict(base); model.split.update(c); t = step_ms()
    print(f"{json.dumps(c):44s} {t:.4f} ms vs heuristic {b:.4f} ms  ({(b / t - 1) * 100:+.2f} %)", flush=True)
