#!/bin/bash
# Same-box comparison of several builds of the library: bench.py decode line under each, alternating.
#   tools/ab_multi.sh "<bench args>" lib1.so lib2.so ...    ("-" = the in-tree library)
ARGS=$1; shift
for rep in 1 2 3; do
  for lib in "$@"; do
    if [ "$lib" != "-" ]; then export EXL3_HIP_LIB=$lib; else unset EXL3_HIP_LIB; fi
    timeout 300 python bench.py --no-extra --no-cpu $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'], d.get('repeat_ms_per_step'))"
  done
done
