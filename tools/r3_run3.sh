#!/bin/bash
# Round 3, third GPU pass: two units of weight rows in flight (generation 4) against one; timelines; kernel stats of the rotated pipeline. -> gpurun_out/r3d/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3d; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q --maxfail=10 -k "gemv4 or gemm or path" > $O/pytest.txt 2>&1; echo "tests rc=$?" | tee -a $O/pytest.txt
tail -4 $O/pytest.txt
for act in 1 0; do
  echo "== timeline ACT_IN_GEMV=$act (2 units in flight)"
  EXL3_HIP_ACT_IN_GEMV=$act EXL3_HIP_LIB=$R/build/libexl3_hip_timing4.so timeout 300 python tools/gemv_timeline.py 2>&1 | tail -4
done | tee $O/timeline.txt
line() { python -c "
import sys,json
try:
    d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d.get('roofline') or {}
    print('$1', d['value'], d['ms_per_step'], d.get('repeat_ms_per_step'), 'gemv_us', r.get('avg_launch_us'), 'frac', r.get('frac'))
except Exception as e: print('$1', 'FAILED', e)
"; }
for rep in 1 2 3; do
  EXL3_HIP_LIB=$R/build/libexl3_hip_pfu1.so EXL3_HIP_ACT_IN_GEMV=0 timeout 300 python bench.py --no-extra --no-cpu --no-prefill 2>/dev/null | line pfu1_glue_act
  EXL3_HIP_ACT_IN_GEMV=0 timeout 300 python bench.py --no-extra --no-cpu --no-prefill 2>/dev/null | line pfu2_glue_act
  EXL3_HIP_LIB=$R/build/libexl3_hip_pfu1.so timeout 300 python bench.py --no-extra --no-cpu --no-prefill 2>/dev/null | line pfu1_act_in_gemv
  timeout 300 python bench.py --no-extra --no-cpu --no-prefill 2>/dev/null | line pfu2_act_in_gemv
done 2>&1 | tee $O/ab.txt
EXL3_HIP_ROTATE_ABOVE=0 timeout 300 python bench.py --no-extra --no-cpu --no-prefill 2>/dev/null | line pfu2_rotated | tee -a $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for cfg in "rot:EXL3_HIP_ROTATE_ABOVE=0" "glue:EXL3_HIP_ACT_IN_GEMV=0"; do
  n=${cfg%%:*}; v=${cfg##*:}
  env $v timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$n -o out --output-format csv -- python $R/bench.py --no-extra --no-cpu --no-prefill --steps 20 > $O/prof_$n.json 2> $O/prof_$n.err
  cp $O/prof_$n/out_kernel_stats.csv $O/kernel_stats_$n.csv 2>/dev/null; rm -rf $O/prof_$n
  echo "== $n"; head -9 $O/kernel_stats_$n.csv | cut -c1-140
done
