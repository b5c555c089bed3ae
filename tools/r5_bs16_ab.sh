#!/bin/bash
# bs-16 decode step (C++ harness, folded glue pipeline) for the main library and alternates: tools/r5_bs16_ab.sh <tag> [alt dirs...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5; mkdir -p $O; T=$1; shift
for rep in 1 2; do
for d in exllamav3_amd "$@"; do
  n=$(basename $d)
  LD_LIBRARY_PATH=$PWD/$d timeout 120 tools/bin/decode_step_harness 32 3 0 8 0 0 1 16 > $O/${T}_bs16_${n}_$rep.json 2> $O/${T}_bs16_${n}_$rep.err
  echo "$n rep $rep: $(grep -o '"ms_per_step": [0-9.]*, "tok_s": [0-9.]*' $O/${T}_bs16_${n}_$rep.json) $(grep -o '"nonfinite": [0-9]*' $O/${T}_bs16_${n}_$rep.json) $(tail -c 200 $O/${T}_bs16_${n}_$rep.err)"
done; done
EXL3_HIP_LIB=$PWD/${1:-exllamav3_amd}/libexl3_hip.so timeout 300 python -m pytest tests/test_gpu_gemm.py -q -m gpu -k "gemm3 or batch_sizes" 2>&1 | tail -3
