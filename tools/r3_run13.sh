#!/bin/bash
# same-box A/B of two library builds on the prefill chunk: build/libexl3_hip_old.so vs build/libexl3_hip_new.so
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_gemm.py -q -x -k "reconstruct or prefill or had" 2>&1 | tail -2
for rep in 1 2 3; do for v in old new; do
  EXL3_HIP_LIB=$R/build/libexl3_hip_$v.so timeout 300 python bench.py --no-extra --no-cpu --steps 10 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); p = d.get('prefill') or {}
print('$v', 'decode', d['value'], 'prefill', p.get('value'), (p.get('roofline') or {}).get('frac'))"
done; done | tee $O/ab.txt
