#!/usr/bin/env python3
"""Sweep the k-split / waves-per-workgroup knobs of the deferred-epilogue GEMV launches of one Llama-3.1-8B layer
(hipGraph of all 32 layers' instances of a call type = cold weights; GPU-side event timing)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext
from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama

ap = argparse.ArgumentParser()
ap.add_argument("--wgs", default="0,1,2,3,4,6,8")
ap.add_argument("--waves", default="0")
ap.add_argument("--cb", type=int, default=2)
ap.add_argument("--batch", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0"); ext.init(0)
model = SyntheticEXL3Llama(SHAPES["llama-3.1-8b"], K=4, cb=a.cb, device=dev)
model.alloc_state(a.batch)
model.decode_step_fused(); torch.cuda.synchronize()
names = ["qkv", "o", "gate_up", "down"]
for mw in [int(v) for v in a.waves.split(",")]:
    for wg in [int(v) for v in a.wgs.split(",")]:
        ext._lib.lib().exl3_set_gemv_defer_wg_per_cu(wg); ext.set_gemv_max_waves(mw)
        calls = model.gemv_calls(True)
        res = {}
        st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for i, nm in enumerate(names):
                grp = calls[i:len(calls) - 1:4]
                for c in grp: c()
                st.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    for c in grp: c()
                g.replay(); st.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(5): g.replay()
                e1.record(st); st.synchronize()
                res[nm] = round(e0.elapsed_time(e1) * 1e3 / 5 / len(grp), 2)
        res["sum"] = round(sum(res.values()), 2)
        print(json.dumps({"wg_per_cu": wg, "max_waves": mw, **res}), flush=True)
