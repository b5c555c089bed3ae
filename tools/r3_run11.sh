#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3k; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_attn_prefill.py -q -x > $O/t.log 2>&1; tail -2 $O/t.log
for rep in 1 2; do for cfg in "4096 0" "2048 0" "1024 8192" "8192 0"; do timeout 120 python tools/bench_attn_prefill.py $cfg 2>/dev/null | tail -1; done; done | tee $O/bench.txt
