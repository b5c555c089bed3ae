#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3p; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_gemv4.py tests/test_gpu_path.py tests/test_gpu_gemm.py -q -x > $O/t1.log 2>&1; tail -2 $O/t1.log
for rep in 1 2 3; do for v in old new; do
  L=$R/build/libexl3_hip_old.so; [ $v = new ] && L=$R/exllamav3_amd/libexl3_hip.so
  EXL3_HIP_LIB=$L timeout 300 python bench.py --no-prefill --no-extra --no-cpu 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v bs1', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('avg_launch_us'))"
done; done | tee $O/ab.txt
