#!/bin/bash
# short profile refresh: rocprofv3 kernel stats of the bs-1, Mixtral and prefill bench configs (no PMC passes)
TAG=r03
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
run() { n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$n -o out --output-format csv -- python $R/bench.py --no-extra --no-cpu "$@" > $O/${TAG}_bench_${n}_under_rocprof.json 2> $O/$n.err
  cp $O/$n/out_kernel_stats.csv $O/${TAG}_bench_${n}_kernel_stats.csv 2>/dev/null; rm -rf $O/$n; }
run bs1 --no-prefill --steps 20
run bs16 --batch 16 --no-prefill --steps 20
run mixtral --model mixtral-8x7b --steps 10
run prefill --steps 5
ls $O | grep r03 | head -20
