#!/bin/bash
# Round 3, second GPU pass: GPU suite, generation-4 phase timelines, A/B of the pipelines, split / wave-count sweeps.  -> gpurun_out/r3b/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=30 > $O/pytest_all.txt 2>&1; echo "all tests rc=$?" | tee -a $O/pytest_all.txt
tail -8 $O/pytest_all.txt
for act in 1 0; do
  echo "== timeline ACT_IN_GEMV=$act"
  EXL3_HIP_ACT_IN_GEMV=$act EXL3_HIP_LIB=$R/build/libexl3_hip_timing4.so timeout 300 python tools/gemv_timeline.py 2>&1 | tail -6
done | tee $O/timeline.txt
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
for act in 0 1; do
  echo "== split sweep ACT_IN_GEMV=$act"
  EXL3_HIP_ACT_IN_GEMV=$act CANDS='{"qkv":[4,8,16,32],"o":[8,16,32],"gu":[4,8,16],"down":[8,14,16,28]}' timeout 400 python tools/sweep_split.py 16 2>&1 | tail -20
done | tee $O/sweep.txt
for mw in 2 4 8; do
  echo "== max waves $mw (ACT_IN_GEMV=0)"
  EXL3_HIP_ACT_IN_GEMV=0 MAXW=$mw CANDS='{}' timeout 300 python tools/sweep_split.py 16 2>&1 | tail -2
done | tee $O/maxw.txt
