#!/bin/bash
# persistent step, same-box A/B of library builds: tools/r5_pstep_ab.sh <libdir A> <libdir B> ... (alternating, 2 rounds, 8b and 1b); gate on the first
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5; mkdir -p $O
H=tools/bin/pstep_harness
export LD_LIBRARY_PATH=$PWD/$1:$LD_LIBRARY_PATH
for L in "$@"; do
m=1b
LD_LIBRARY_PATH=$PWD/$L:$LD_LIBRARY_PATH H_SPIN_LIMIT=3000 timeout 60 $H $m 2 1 "3" > $O/g_${m}_l2.json 2> $O/g_${m}_l2.err; echo "rc=$?" >> $O/g_${m}_l2.err
python3 - <<PY || { echo "GATE FAILED ($L): see $O/g_${m}_l2.json"; tail -c 1500 $O/g_${m}_l2.json; tail -c 300 $O/g_${m}_l2.err; exit 0; }
import re,sys
s=open("$O/g_${m}_l2.json").read()
bad = '"edge_timeout": 1' in s or 'edge_timeout_during_replays": 1' in s or not '"best"' in s
d=[float(x) for x in re.findall(r'"rms_diff": ([0-9.e+-]+)', s)]
print("gate $L rms_diff", d)
sys.exit(1 if bad or max(d+[1.0] if not d else d) > 0.005 else 0)
PY
done
for r in $(seq 1 ${ROUNDS:-3}); do for L in "$@"; do for m in 8b 1b; do
  LD_LIBRARY_PATH=$PWD/$L:$LD_LIBRARY_PATH H_SPIN_LIMIT=20000 timeout 90 $H $m 0 2 "3" > $O/ab_${m}.json 2> $O/ab_${m}.err
  python3 - <<PY
import re
s=open("$O/ab_${m}.json").read()
print("$L $m", re.findall(r'"variant": "([a-z0-9_]+)", "ms_per_step": ([0-9.]+)', s[s.find('"best"'):]))
PY
done; done; done
