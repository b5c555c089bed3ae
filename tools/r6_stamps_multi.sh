#!/bin/bash
# 8B phase stamps of several library builds: tools/r6_stamps_multi.sh <libdir> ...  -> the gate|up / down / head lines of each
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6; mkdir -p $O
H=tools/bin/pstep_harness
for L in "$@"; do
  T=$(basename $L)
  LD_LIBRARY_PATH=$PWD/$L:$LD_LIBRARY_PATH H_SPIN_LIMIT=20000 timeout 90 $H 8b 0 1 "3" $O/s_8b_$T.bin 2>&1 | grep -o '"best".*'
  echo "== $T"; python3 tools/pstep_stamps.py $O/s_8b_$T.bin 32 $O/stamps_8b_$T.json | grep -B1 "period\|head"
  rm -f $O/s_8b_$T.bin
done
