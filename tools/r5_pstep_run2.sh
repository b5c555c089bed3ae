#!/bin/bash
# round 5: persistent decode step vs the launch-per-op pipeline (C++ harness).  A 2-layer run with a short spin limit goes first: a broken kernel costs seconds, not minutes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5; mkdir -p $O
T=${1:-c}
export LD_LIBRARY_PATH=$PWD/exllamav3_amd:$LD_LIBRARY_PATH
H=tools/bin/pstep_harness
H_SPIN_LIMIT=3000 timeout 60 $H 8b 2 1 "0,3" > $O/${T}_8b_l2.json 2> $O/${T}_8b_l2.err; echo "rc=$?" >> $O/${T}_8b_l2.err
python3 - <<PY || { echo "GATE FAILED: see $O/${T}_8b_l2.json"; tail -c 1500 $O/${T}_8b_l2.json; tail -c 300 $O/${T}_8b_l2.err; exit 0; }
import re,sys
s=open("$O/${T}_8b_l2.json").read()
bad = '"edge_timeout": 1' in s or 'edge_timeout_during_replays": 1' in s or not '"best"' in s
d=[float(x) for x in re.findall(r'"rms_diff": ([0-9.e+-]+)', s)]
sys.exit(1 if bad or max(d+[1.0] if not d else d) > 0.005 else 0)
PY
H_SPIN_LIMIT=20000 timeout 90 $H 8b 0 3 "3,2" $O/${T}_8b_stamps.bin > $O/${T}_8b.json 2> $O/${T}_8b.err; echo "rc=$?" >> $O/${T}_8b.err
H_SPIN_LIMIT=20000 timeout 90 $H 1b 0 3 "3,2" $O/${T}_1b_stamps.bin > $O/${T}_1b.json 2> $O/${T}_1b.err; echo "rc=$?" >> $O/${T}_1b.err
for m in 8b 1b; do python3 - <<PY
s=open("$O/${T}_$m.json").read()
i=s.find('"vs_baseline"'); j=s.find('"ms_per_step"'); print("$m", s[i:j][:700])
k=s.find('"best"'); print(s[k:])
PY
tail -c 200 $O/${T}_$m.err; done
