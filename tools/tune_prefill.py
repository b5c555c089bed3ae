#!/usr/bin/env python3
"""Prefill chunk time with the heuristic-list GEMM autotune (default) vs the exhaustive one (EXL3_HIP_HGEMM_TUNE=2), 4 layers."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext
from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama
dev = torch.device("cuda:0"); ext.init(0)
model = SyntheticEXL3Llama(SHAPES["llama-3.1-8b"], K=4, cb=2, device=dev, layers=4)
t0 = time.perf_counter(); model.prefill_chunk(4096); torch.cuda.synchronize(); print(f"first chunk (tuning): {time.perf_counter() - t0:.1f} s", flush=True)
for _ in range(2):
    t0 = time.perf_counter(); model.prefill_chunk(4096); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"EXL3_HIP_HGEMM_TUNE={os.environ.get('EXL3_HIP_HGEMM_TUNE', '1')}: {dt * 1e3 / 4:.3f} ms per layer -> {4096 / (dt / 4 * 32):.0f} tok/s at 32 layers", flush=True)
