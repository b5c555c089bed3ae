#!/bin/bash
# prefill attention: two-group kernel vs the round-2 kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3j; mkdir -p $O; cd $R
EXL3_HIP_ATTN_PP=2 timeout 600 python -m pytest tests/test_gpu_attn_prefill.py -q -x > $O/t_pp2.log 2>&1; tail -3 $O/t_pp2.log
timeout 600 python -m pytest tests/test_gpu_fullsize.py -q -x -k attn > $O/t_full.log 2>&1; tail -2 $O/t_full.log
for pp in 0 1 0 1; do
  for cfg in "4096 0" "2048 0" "1024 8192" "8192 0"; do
    echo -n "pp=$pp "; EXL3_HIP_ATTN_PP=$pp timeout 120 python tools/bench_attn_prefill.py $cfg 2>/dev/null | tail -1
  done
done | tee $O/bench.txt
