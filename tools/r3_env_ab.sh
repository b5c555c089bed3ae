#!/bin/bash
# same-box A/B of an environment toggle on the default bs-1 decode step: tools/r3_env_ab.sh VAR=a VAR=b [more bench args]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3s; mkdir -p $O; cd $R
A=$1; B=$2; shift 2
for rep in 1 2 3; do for v in "$A" "$B"; do
  env $v timeout 300 python bench.py --no-cpu --steps 60 --warmup 10 "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])"
done; done | tee $O/ab.txt
