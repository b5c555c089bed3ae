#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3n; mkdir -p $O; cd $R
for rep in 1 2; do for cb in 22528 36864; do for b in 16 8; do
  EXL3_HIP_GEMM3_CHUNK_BYTES=$cb timeout 300 python bench.py --batch $b --no-prefill --no-extra --no-cpu --steps 30 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('chunk$cb bs$b', d['value'], d['ms_per_step'])"
done; done; done | tee $O/ab2.txt
