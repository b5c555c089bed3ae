#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/exllamav3_amd:$LD_LIBRARY_PATH
H_SPIN_LIMIT=3000 timeout 60 tools/bin/pstep_harness 1b 0 2 "2" > $O/r_1b.json 2> $O/r_1b.err; grep -o '"best".*' $O/r_1b.json | cut -c1-330
timeout 600 python -m pytest tests/test_gpu_path.py -q -m gpu -x -k "attention" 2>&1 | tail -3
timeout 300 python -m pytest tests/test_gpu_pstep.py -q -m gpu -x -k "matches_oracle_and_fx" 2>&1 | tail -2
timeout 300 python tools/bench_decode_ctx.py 1000 4000 16000 2>&1 | tail -5
