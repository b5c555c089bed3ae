#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3h; mkdir -p $O; cd $R
timeout 400 python -m pytest tests/test_gpu_prefill.py -m gpu -q 2>&1 | tail -6 | cut -c1-300
timeout 300 python tools/bench_gemm_nt.py 2>&1 | tail -5
for v in 0 1; do
  EXL3_HIP_GEMM_NT=$v timeout 300 python bench.py --no-extra --no-cpu --steps 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); p=d['prefill']
print('EXL3_HIP_GEMM_NT=$v prefill', p['value'], p['ms_per_chunk'], p['roofline']['frac'], 'with_attention', p['with_attention']['value'])
"
done | tee $O/prefill_ab.txt
