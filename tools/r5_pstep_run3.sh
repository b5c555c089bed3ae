#!/bin/bash
# persistent step: DIRECT residual edges vs owners (EXL3_HIP_PSTEP_OWNERS=1), C++ harness; a 2-layer gate with a short spin limit first
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/exllamav3_amd:$LD_LIBRARY_PATH
H=tools/bin/pstep_harness
for m in 8b 1b; do
H_SPIN_LIMIT=3000 timeout 60 $H $m 2 1 "3" > $O/g_${m}_l2.json 2> $O/g_${m}_l2.err; echo "rc=$?" >> $O/g_${m}_l2.err
python3 - <<PY || { echo "GATE FAILED ($m): see $O/g_${m}_l2.json"; tail -c 1500 $O/g_${m}_l2.json; tail -c 300 $O/g_${m}_l2.err; exit 0; }
import re,sys
s=open("$O/g_${m}_l2.json").read()
bad = '"edge_timeout": 1' in s or 'edge_timeout_during_replays": 1' in s or not '"best"' in s
d=[float(x) for x in re.findall(r'"rms_diff": ([0-9.e+-]+)', s)]
print("gate $m rms_diff", d)
sys.exit(1 if bad or max(d+[1.0] if not d else d) > 0.005 else 0)
PY
done
for r in 1 2; do for own in 0 1; do for m in 8b 1b; do
  EXL3_HIP_PSTEP_OWNERS=$own H_SPIN_LIMIT=20000 timeout 90 $H $m 0 3 "3" > $O/o${own}_${m}.json 2> $O/o${own}_${m}.err
  python3 - <<PY
s=open("$O/o${own}_${m}.json").read()
k=s.find('"best"'); i=s.find('"plan"'); print("owners=$own $m", s[k:].strip()[:400]); 
if $r == 1: print("   ", s[i:i+420].split('\n')[0])
PY
done; done; done
H_SPIN_LIMIT=20000 timeout 90 $H 1b 0 1 "3" $O/dir_1b_stamps.bin > /dev/null 2>&1
H_SPIN_LIMIT=20000 timeout 90 $H 8b 0 1 "3" $O/dir_8b_stamps.bin > /dev/null 2>&1
