#!/bin/bash
# Round 3, first GPU pass: the GPU suite, then same-box A/B of the bs-1 decode line under the generation-4 GEMV and its pipeline choices.
#   gpurun -- 'bash tools/r3_ab_gen4.sh'   -> gpurun_out/r3a/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3a; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 -x -k "gemv4 or attn_decode or routing or moe" > $O/pytest_new.txt 2>&1; echo "new tests rc=$?" | tee -a $O/pytest_new.txt
tail -5 $O/pytest_new.txt
line() { python -c "
import sys,json
try:
    d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d.get('roofline') or {}
    print('$1', d['value'], d['ms_per_step'], d.get('repeat_ms_per_step'), 'gemv_us', r.get('avg_launch_us'), 'frac', r.get('frac'))
except Exception as e: print('$1', 'FAILED', e)
"; }
for rep in 1 2; do
  EXL3_HIP_GEMV_GEN4=0 timeout 300 python bench.py --no-extra --no-cpu --no-prefill 2>$O/err_gen2.txt | line gen2
  EXL3_HIP_GEMV_GEN4=1 timeout 300 python bench.py --no-extra --no-cpu --no-prefill 2>$O/err_g4act.txt | line gen4_act_in_gemv
  EXL3_HIP_GEMV_GEN4=1 EXL3_HIP_ACT_IN_GEMV=0 timeout 300 python bench.py --no-extra --no-cpu --no-prefill 2>$O/err_g4glue.txt | line gen4_glue_act
  EXL3_HIP_GEMV_GEN4=1 EXL3_HIP_ROTATE_ABOVE=0 timeout 300 python bench.py --no-extra --no-cpu --no-prefill 2>$O/err_g4rot.txt | line gen4_rotated_pipeline
done 2>&1 | tee $O/ab.txt
# kernel-level view of the two generation-4 pipelines
cd /tmp && export TMPDIR=/tmp
for cfg in "act:1" "glue:0"; do
  n=${cfg%%:*}; v=${cfg##*:}
  EXL3_HIP_ACT_IN_GEMV=$v timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$n -o out --output-format csv -- python $R/bench.py --no-extra --no-cpu --no-prefill --steps 20 > $O/prof_$n.json 2> $O/prof_$n.err
  cp $O/prof_$n/out_kernel_stats.csv $O/kernel_stats_$n.csv 2>/dev/null; rm -rf $O/prof_$n
  head -8 $O/kernel_stats_$n.csv | cut -c1-160
done
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 > $O/pytest_all.txt 2>&1; echo "all tests rc=$?" | tee -a $O/pytest_all.txt
tail -15 $O/pytest_all.txt
