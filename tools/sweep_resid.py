"""Whole-step time of decode_step_resid (hipGraph replay) vs the split-k factor of its call types, and per-kernel times under
rocprofv3 when PROF=1.  env: BSZ, CANDS (json), STEP=fused|resid."""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama

dev = torch.device("cuda:0")
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model = SyntheticEXL3Llama(SHAPES["llama-3.1-8b"], K=4, cb=2, device=dev, kv_bits=4, layers=layers)
model.alloc_state(int(os.environ.get('BSZ', '1')))
which = os.environ.get("STEP", "resid")
fn = model.decode_step_resid if which == "resid" else model.decode_step_fused


def step_us():
    fn(); torch.cuda.synchronize()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            fn()
        g.replay(); st.synchronize()
        best = 1e9
        for _ in range(3):
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10): g.replay()
            e1.record(st); st.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 10)
    return best / layers


if os.environ.get("PROF"):
    fn(); torch.cuda.synchronize()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    sys.exit(0)
base = step_us()
print(f"{which} defaults: {base:.2f} us/layer")
cands = {"qkv_resid": [(2, 16), (4, 8), (4, 16), (4, 32), (8, 32)], "o_resid": [(0, 8), (0, 16), (0, 32), (4, 32)],
         "gu_resid": [(7, 8), (7, 16), (8, 16), (14, 16), (16, 16)], "down_resid": [(0, 8), (0, 16), (0, 28), (4, 28)]}
if os.environ.get("CANDS"): cands = json.loads(os.environ["CANDS"])
for key, vals in cands.items():
    for v in vals:
        model.split = dict(SyntheticEXL3Llama.split); model.split[key] = tuple(v)
        try:
            print(f"{key:10s} cpw,S={tuple(v)}: {step_us():7.2f} us/layer")
        except Exception as e:
            print(f"{key:10s} cpw,S={tuple(v)}: failed {type(e).__name__}: {str(e)[:80]}")
    model.split = dict(SyntheticEXL3Llama.split)
