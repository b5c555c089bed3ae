#!/bin/bash
# round 4: generation-3 (5..64 rows) with 4 / 8 / 16-wave workgroups -- same-box A/B in the C++ decode-step harness at batch 16 (no Python on the box)
H=./tools/bin/decode_step_harness
O=gpurun_out/r4c; mkdir -p $O
run() { LD_LIBRARY_PATH=build/lite_$1 EXL3_HIP_GEMM3_CPW=$2 $H 32 3 $3 0 $4 $5 1 16 2>>$O/err.log | tail -1 | sed -e 's/"model.*pipeline via the C ABI", //'; }
for rep in 1 2; do for v in "base 0" "g3w 0" "g3w 1" "g3w 2" "g3w 4"; do echo -n "$v "; run $v 0 0 0; done; done | tee $O/ab.txt
echo "# split sweep (qkv gate_up down) at forced column blocks per workgroup" | tee -a $O/ab.txt
for c in 4 2; do for s in "0 2 0" "0 4 0" "0 8 0" "0 4 14" "0 4 16" "0 4 28" "4 4 0" "8 4 0" "16 4 0"; do echo -n "g3w cpw=$c [$s] "; run g3w $c $s; done; done | tee -a $O/ab.txt
