#!/usr/bin/env python3
"""One prefill chunk of the Llama-3.1-8B hot path (few layers) for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exllamav3_amd import ext
from exllamav3_amd.llama_path import SHAPES, SyntheticEXL3Llama
dev = torch.device("cuda:0"); ext.init(0)
model = SyntheticEXL3Llama(SHAPES["llama-3.1-8b"], K=4, cb=2, device=dev, layers=4)
model.prefill_chunk(4096); torch.cuda.synchronize()
model.prefill_chunk(4096); torch.cuda.synchronize()
print("done")
