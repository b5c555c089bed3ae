#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6; mkdir -p $O
L=${LIBDIR:-exllamav3_amd}
LD_LIBRARY_PATH=$PWD/$L:$LD_LIBRARY_PATH H_SPIN_LIMIT=20000 timeout 90 tools/bin/pstep_harness 8b 0 1 "3" $O/s_str.bin 2>&1 | grep -o '"best".*' | cut -c1-200
python3 tools/pstep_stragglers.py $O/s_str.bin 32; rm -f $O/s_str.bin
