#!/bin/bash
# round 3, run 7: table modes of the generation-4 GEMV + Mixtral fx pipeline: tests, then A/B on the Mixtral step
mkdir -p gpurun_out/r3g
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gemv4.py tests/test_gpu_gemm.py -q -x -k "table or routing_on or indexed or moe" > gpurun_out/r3g/t1.log 2>&1; echo "t1 rc=$?" >> gpurun_out/r3g/t1.log
timeout 900 python -m pytest tests/test_gpu_path.py tests/test_gpu_fullsize.py -q -x -k "mixtral or moe or Mixtral" > gpurun_out/r3g/t2.log 2>&1; echo "t2 rc=$?" >> gpurun_out/r3g/t2.log
tail -5 gpurun_out/r3g/t1.log gpurun_out/r3g/t2.log
for rep in 1 2; do
  for p in glue fx; do
    timeout 600 python bench.py --model mixtral-8x7b --pipeline $p --no-cpu --steps 30 --warmup 5 > gpurun_out/r3g/mix_${p}_$rep.json 2> gpurun_out/r3g/mix_${p}_$rep.err
    python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r3g/mix_${p}_$rep.json").read().strip().splitlines()[-1]); print("$p $rep", d["value"], d["ms_per_step"])
except Exception as e: print("$p $rep failed", e)
PY
  done
done
