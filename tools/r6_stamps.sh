#!/bin/bash
# phase stamps of the persistent step (C++ harness): tools/r6_stamps.sh [tag]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-cur}
O=gpurun_out/r6; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/${LIBDIR:-exllamav3_amd}:$LD_LIBRARY_PATH
H=tools/bin/pstep_harness
H_SPIN_LIMIT=20000 timeout 90 $H 1b 0 1 "3" $O/s_1b_$T.bin 2>&1 | grep -o '"best".*'
H_SPIN_LIMIT=20000 timeout 90 $H 8b 0 1 "3" $O/s_8b_$T.bin 2>&1 | grep -o '"best".*'
python3 tools/pstep_stamps.py $O/s_1b_$T.bin 16 $O/stamps_1b_$T.json
python3 tools/pstep_stamps.py $O/s_8b_$T.bin 32 $O/stamps_8b_$T.json
rm -f $O/s_1b_$T.bin $O/s_8b_$T.bin
