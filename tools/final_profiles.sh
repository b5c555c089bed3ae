# Round-end profile refresh: rocprofv3 kernel stats of the decode bench at bs 1 and bs 16 (every step under its own timeout).
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
timeout 150 rocprofv3 --kernel-trace --stats -d $O/bs1 -o out --output-format csv -- python $R/bench.py --no-prefill --no-cpu --steps 20 > $O/bs1.log 2>&1
timeout 150 rocprofv3 --kernel-trace --stats -d $O/bs16 -o out --output-format csv -- python $R/bench.py --batch 16 --no-prefill --no-cpu --steps 20 > $O/bs16.log 2>&1
rm -f $O/*/out_kernel_trace.csv
ls $O $O/bs1 $O/bs16; grep "^{" $O/bs1.log | cut -c1-160; grep "^{" $O/bs16.log | cut -c1-160
