#!/bin/bash
# Mixtral fx: k-split sweep of the block's two indexed launches
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3i; mkdir -p $O; cd $R
for sp in 0,14 0,16 0,19 0,23 0,16 0,14; do
  EXL3_HIP_MOE_FX_SPLIT=$sp timeout 300 python bench.py --model mixtral-8x7b --pipeline fx --no-cpu --steps 30 --warmup 5 2>$O/err_$sp.txt | python -c "
import sys,json
try:
    d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$sp', d['value'], d['ms_per_step'])
except Exception as e: print('$sp', 'FAILED', e)
"
done 2>&1 | tee $O/sweep.txt
