#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3r; mkdir -p $O; cd $R
for rep in 1 2; do for v in base noprep; do
  L=$R/exllamav3_amd/libexl3_hip.so; [ $v != base ] && L=$R/build/libexl3_hip_$v.so
  EXL3_HIP_LIB=$L timeout 300 python bench.py --no-prefill --no-extra --no-cpu 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v bs1', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('avg_launch_us'))"
done; done | tee $O/ab.txt
