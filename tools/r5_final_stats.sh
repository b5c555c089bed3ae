#!/bin/bash
# kernel stats of the persistent step after the last kernel change of round 5 (8B and 1B, rocprofv3 --kernel-trace --stats around the short bench line): gpurun_out/final/
TAG=r05b
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { n=$1; shift
  timeout 70 rocprofv3 --kernel-trace --stats -d $O/$n -o out --output-format csv -- python $R/bench.py --no-extra --no-cpu "$@" > $O/${TAG}_bench_${n}_under_rocprof.json 2> $O/$n.err
  cp $O/$n/out_kernel_stats.csv $O/${TAG}_bench_${n}_kernel_stats.csv 2>/dev/null; rm -rf $O/$n; }
run bs1 --no-prefill --steps 20
run 1b_persistent --model llama-3.2-1b --no-prefill --steps 20
head -4 $O/${TAG}_bench_bs1_kernel_stats.csv; head -3 $O/${TAG}_bench_1b_persistent_kernel_stats.csv
