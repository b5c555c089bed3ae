#!/bin/bash
# Round 3, fourth GPU pass: the fixed-point residual ("fx") pipeline: tests, A/B against the glue pipeline, kernel stats. -> gpurun_out/r3e/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3e; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q --maxfail=10 -k "fixed_point or gemv4 or resid_in_gemv" > $O/pytest.txt 2>&1; echo "tests rc=$?" | tee -a $O/pytest.txt
tail -25 $O/pytest.txt | cut -c1-300
line() { python -c "
import sys,json
try:
    d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d.get('roofline') or {}
    print('$1', d['value'], d['ms_per_step'], d.get('repeat_ms_per_step'), 'gemv_us', r.get('avg_launch_us'), 'frac', r.get('frac'))
except Exception as e: print('$1', 'FAILED', e)
"; }
for rep in 1 2 3; do
  EXL3_HIP_ACT_IN_GEMV=0 timeout 300 python bench.py --no-extra --no-cpu --no-prefill 2>$O/err_glue.txt | line glue_pipeline_glue_act
  timeout 300 python bench.py --no-extra --no-cpu --no-prefill --pipeline fx 2>$O/err_fx.txt | line fx_pipeline
done 2>&1 | tee $O/ab.txt
tail -5 $O/err_fx.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_fx -o out --output-format csv -- python $R/bench.py --no-extra --no-cpu --no-prefill --steps 20 --pipeline fx > $O/prof_fx.json 2> $O/prof_fx.err
cp $O/prof_fx/out_kernel_stats.csv $O/kernel_stats_fx.csv 2>/dev/null; rm -rf $O/prof_fx
head -10 $O/kernel_stats_fx.csv | cut -c1-140
