#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3n; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_fullsize.py tests/test_gpu_path.py -q -x -k "gemm3 or generation_3 or batch or rows or fullsize or llama or fused" > $O/t1.log 2>&1; tail -2 $O/t1.log
for rep in 1 2; do for v in old new; do for b in 16 8; do
  L=$R/build/libexl3_hip_old.so; [ $v = new ] && L=$R/exllamav3_amd/libexl3_hip.so
  EXL3_HIP_LIB=$L timeout 300 python bench.py --batch $b --no-prefill --no-extra --no-cpu --steps 30 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v bs$b', d['value'], d['ms_per_step'])"
done; done; done | tee $O/ab5.txt
EXL3_HIP_LIB=$R/build/libexl3_hip_timing.so BSZ=16 timeout 300 python tools/gemv_timeline.py 2>/dev/null | grep "^qkv\|^o \|^gate\|^down" | python -c "
import sys, json
for l in sys.stdin:
    nm, js = l.split(' ', 1); d = json.loads(js); print(nm, d['workgroups'], d['per_wg_phase_us_p50_p90'])"
