"""Whole-step time of the glue pipeline (hipGraph replay) as a function of the split-k factor of each call type.  env: BSZ, G3MIN, CANDS (json)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama

dev = torch.device("cuda:0")
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model = SyntheticEXL3Llama(SHAPES[os.environ.get("MODEL", "llama-3.1-8b")], K=4, cb=2, device=dev, kv_bits=4, layers=layers)
model.alloc_state(int(os.environ.get('BSZ', '1')))
from exllamav3_amd import ext
if os.environ.get('G3MIN'): ext.set_gemm3_min_rows(int(os.environ['G3MIN']))
if os.environ.get('MAXW'): ext.set_gemv_max_waves(int(os.environ['MAXW']))

STEP = getattr(model, os.environ.get('STEP', 'decode_step_fused'))

def step_us():
    STEP(); torch.cuda.synchronize()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            STEP()
        g.replay(); st.synchronize()
        best = 1e9
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10): g.replay()
            e1.record(st); st.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 10)
    return best / layers

base = step_us()
print(f"heuristic: {base:.2f} us/layer")
cands = {"qkv": [16, 32], "o": [16, 32], "gu": [2, 4, 8, 16, 32], "down": [16, 28, 56]}
if os.environ.get("CANDS"): cands = json.loads(os.environ["CANDS"])
for key, vals in cands.items():
    for v in vals:
        model.split = dict(SyntheticEXL3Llama.split); model.split[key] = v
        print(f"{key:5s} S={v:3d}: {step_us():7.2f} us/layer")
    model.split = dict(SyntheticEXL3Llama.split)
