#!/bin/bash
# quick correctness gate of the persistent step for several library variants: tools/r5_gate.sh <tag> [alt dirs...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5; mkdir -p $O; T=$1; shift
for d in exllamav3_amd "$@"; do
  n=$(basename $d)
  LD_LIBRARY_PATH=$PWD/$d H_SPIN_LIMIT=3000 timeout 60 tools/bin/pstep_harness 8b 2 1 "0,2" > $O/${T}_gate_$n.json 2> $O/${T}_gate_$n.err; echo "rc=$?" >> $O/${T}_gate_$n.err
  echo "== $n"; python3 - <<PY
import re
s=open("$O/${T}_gate_$n.json").read()
print(re.findall(r'"variant": "[a-z_0-9]+", "logits_rms": [0-9.]+, "max_abs_diff": [0-9.e+-]+, "rms_diff": [0-9.e+-]+', s)); print(re.findall(r'"edge_timeout[a-z_]*": \d', s)); k=s.find('"best"'); print(s[k:k+500])
PY
done
