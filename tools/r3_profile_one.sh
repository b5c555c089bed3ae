#!/bin/bash
# rocprofv3 kernel stats of one bench configuration: tools/r3_profile_one.sh NAME [bench args]
TAG=${TAG:-r04}
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
n=$1; shift
timeout 300 rocprofv3 --kernel-trace --stats -d $O/$n -o out --output-format csv -- python $R/bench.py --no-extra --no-cpu "$@" > $O/${TAG}_bench_${n}_under_rocprof.json 2> $O/$n.err
cp $O/$n/out_kernel_stats.csv $O/${TAG}_bench_${n}_kernel_stats.csv 2>/dev/null; rm -rf $O/$n
head -12 $O/${TAG}_bench_${n}_kernel_stats.csv | cut -c1-150
