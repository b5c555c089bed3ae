#!/bin/bash
# round 4: (column blocks per workgroup, split) sweep of the generation-3 launches in the whole batch-16 step (C++ harness; one process per setting)
H=./tools/bin/decode_step_harness
O=gpurun_out/r4d; mkdir -p $O
run() { LD_LIBRARY_PATH=build/lite_g3w EXL3_HIP_GEMM3_CPW=$1 $H 32 2 $2 0 $3 $4 1 16 2>>$O/err.log | tail -1 | sed -e 's/.*"us_per_layer": \([0-9.]*\).*/\1/'; }
echo "cpw qkv gu down us_per_layer" | tee $O/sweep.txt
for c in 1 2; do
  for q in 4 8 16; do for g in 2 4 8; do for d in 8 14 16 28; do echo "$c $q $g $d $(run $c $q $g $d)"; done; done; done
done | tee -a $O/sweep.txt
sort -k5 -n $O/sweep.txt | head -12
