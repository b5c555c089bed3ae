#!/bin/bash
# Same-box A/B of two builds of the library: bench.py decode line under each, alternating (box-to-box variance is ~2-3 %).
#   tools/ab_bench.sh build/libexl3_hip_base.so [bench args...]
BASE=$1; shift
for rep in 1 2 3; do
  for lib in "$BASE" ""; do
    if [ -n "$lib" ]; then export EXL3_HIP_LIB=$lib; else unset EXL3_HIP_LIB; fi
    timeout 300 python bench.py --no-extra --no-cpu "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-new}', d['value'], d['ms_per_step'], d.get('repeat_ms_per_step'))"
  done
done
