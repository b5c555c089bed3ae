#!/bin/bash
# round 6, F12: generation 2 of the NT GEMM -- parity tests first (bounded), then the four 8B prefill shapes against generation 1 and hipBLASLt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r6; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_prefill.py -x -q -k "hand_written" 2>&1 | tail -15 | tee $O/gemm2_tests.txt
GEN=2 timeout 200 python tools/bench_gemm_nt.py 2>&1 | tail -8 | tee $O/gemm2_bench.txt
GEN=1 timeout 200 python tools/bench_gemm_nt.py 2>&1 | tail -8 | tee $O/gemm1_bench.txt
