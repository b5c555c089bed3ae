import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama
dev = torch.device("cuda:0")
m = SyntheticEXL3Llama(SHAPES["llama-3.1-8b"], K=4, cb=2, device=dev, kv_bits=4)
for ra in (False, True, False, True):
    m.reconstruct_ahead = ra
    m.prefill_chunk(4096); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): m.prefill_chunk(4096)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("reconstruct_ahead", ra, round(dt * 1e3, 2), "ms", round(4096 / dt), "tok/s")
