#!/bin/bash
# persistent step, same-box A/B of library builds (C++ harness, alternating): tools/r6_ab_lib.sh <libdir A> <libdir B> ...   (ROUNDS, MODELS env)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6; mkdir -p $O
H=tools/bin/pstep_harness
for r in $(seq 1 ${ROUNDS:-3}); do for L in "$@"; do for m in ${MODELS:-8b 1b}; do
  LD_LIBRARY_PATH=$PWD/$L:$LD_LIBRARY_PATH H_SPIN_LIMIT=20000 timeout 120 $H $m 0 2 "3" > $O/ab_${m}.json 2> $O/ab_${m}.err
  python3 - <<PY
import re
s=open("$O/ab_${m}.json").read()
print("$L $m", re.findall(r'"variant": "([a-z0-9_]+)", "ms_per_step": ([0-9.]+)', s[s.find('"best"'):]), re.findall(r'"rms_diff": ([0-9.e+-]+)', s)[:2], 'TIMEOUT' if '"edge_timeout": 1' in s else '')
PY
done; done; done
