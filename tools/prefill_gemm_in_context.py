"""Per-shape time of the prefill chunk's GEMM calls IN the chunk (events around every ext.hgemm_nt call of model.prefill_chunk), own route against the library route in
the same process: python tools/prefill_gemm_in_context.py   (8 layers; prints avg us per (m, k, n, accumulate, lda) and the chunk's total)"""
import os, sys, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext, linear
from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama
dev = torch.device("cuda:0"); ext.init(0)
model = SyntheticEXL3Llama(SHAPES["llama-3.1-8b"], K=4, cb=2, device=dev, layers=8)
orig = ext.hgemm_nt
rec = []
def timed(a, bt, c, accumulate=False):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); orig(a, bt, c, accumulate); e1.record()
    rec.append(((a.shape[0], a.shape[1], bt.shape[0], bool(accumulate), a.stride(0)), e0, e1))
for route in (1, 0, 1, 0):
    ext._GEMM_NT_OWN = bool(route)
    ext.hgemm_nt = orig
    model.prefill_chunk(4096); torch.cuda.synchronize()
    ext.hgemm_nt = timed; linear.ext.hgemm_nt = timed
    rec.clear()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); model.prefill_chunk(4096); t1.record(); torch.cuda.synchronize()
    acc = collections.defaultdict(list)
    for key, e0, e1 in rec: acc[key].append(e0.elapsed_time(e1) * 1e3)
    print(json.dumps({"own_route": route, "chunk_ms": round(t0.elapsed_time(t1), 3), "gemm_us": {str(k): round(sum(v) / len(v), 1) for k, v in acc.items()},
                      "gemm_sum_per_layer_us": round(sum(sum(v) for v in acc.values()) / 8, 1)}), flush=True)
    ext.hgemm_nt = orig; linear.ext.hgemm_nt = orig
