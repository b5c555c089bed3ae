#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3q; mkdir -p $O; cd $R
for rep in 1 2 3; do for v in 0 1; do
  EXL3_HIP_RECON_MULTI=$v timeout 300 python bench.py --no-extra --no-cpu --steps 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); p = d.get('prefill') or {}
print('multi=$v prefill', p.get('value'), (p.get('roofline') or {}).get('frac'))"
done; done | tee $O/ab.txt
