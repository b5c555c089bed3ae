#!/bin/bash
# same-box A/B of two library builds on the Mixtral fx step: build/libexl3_hip_old.so vs the in-tree library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3s; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_gemv4.py tests/test_gpu_path.py -q -x -k "act or table or mixtral or fixed_point" > $O/t1.log 2>&1; tail -2 $O/t1.log
for rep in 1 2 3; do for v in old new; do
  L=$R/build/libexl3_hip_old.so; [ $v = new ] && L=$R/exllamav3_amd/libexl3_hip.so
  EXL3_HIP_LIB=$L timeout 300 python bench.py --model mixtral-8x7b --pipeline fx --no-cpu --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$v mixtral', d['value'], d['ms_per_step'])"
done; done | tee $O/ab.txt
