#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/${1:-exllamav3_amd}:$LD_LIBRARY_PATH
H=tools/bin/pstep_harness
H_SPIN_LIMIT=20000 timeout 90 $H 1b 0 1 "3" $O/s_1b_stamps.bin 2>&1 | grep -o '"best".*'
H_SPIN_LIMIT=20000 timeout 90 $H 8b 0 1 "3" $O/s_8b_stamps.bin 2>&1 | grep -o '"best".*'
python3 tools/pstep_stamps.py $O/s_1b_stamps.bin 16
python3 tools/pstep_stamps.py $O/s_8b_stamps.bin 32
