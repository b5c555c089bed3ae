#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3p; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_gemv4.py tests/test_gpu_path.py -q -x > $O/t1.log 2>&1; tail -2 $O/t1.log
for rep in 1 2 3; do for v in 0 1; do
  EXL3_HIP_FX_MERGED_BOUNDARIES=$v timeout 300 python bench.py --no-prefill --no-extra --no-cpu 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('merged=$v bs1', d['value'], d['ms_per_step'])"
done; done | tee $O/ab.txt
