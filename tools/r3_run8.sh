#!/bin/bash
# per-kernel durations of the Mixtral step, glue vs fx pipeline (8 layers)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3h; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for p in glue fx; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$p -o out --output-format csv -- python $R/bench.py --model mixtral-8x7b --layers 8 --pipeline $p --no-cpu --steps 30 --warmup 5 > $O/bench_$p.json 2> $O/bench_$p.err
  cp $O/$p/out_kernel_stats.csv $O/stats_$p.csv; rm -rf $O/$p
done
