#!/bin/bash
# Same-box A/B of decode pipelines: tools/r3_ab_pipelines.sh "<name>:<env assignments>:<bench args>" ...   -> gpurun_out/r3ab/ab.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3ab; mkdir -p $O
cd $R
REPS=${REPS:-3}
for rep in $(seq $REPS); do
  for cfg in "$@"; do
    name=${cfg%%:*}; rest=${cfg#*:}; envs=${rest%%:*}; args=${rest#*:}
    env $envs timeout 300 python bench.py --no-extra --no-cpu --no-prefill $args 2>$O/err_$name.txt | python -c "
import sys,json
try:
    d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d.get('roofline') or {}
    print('$name', d['value'], d['ms_per_step'], d.get('repeat_ms_per_step'), 'gemv_us', r.get('avg_launch_us'), 'frac', r.get('frac'))
except Exception as e: print('$name', 'FAILED', e)
"
  done
done 2>&1 | tee -a $O/ab.txt
for cfg in "$@"; do name=${cfg%%:*}; tail -2 $O/err_$name.txt | cut -c1-300; done
