#!/bin/bash
# split-k sweep of the four GEMV launches of a layer in the whole C++ decode step (one process per setting, same box): tok/s of the three-launch form
H=./tools/bin/decode_step_harness
run() { $H 32 2 $1 $2 $3 $4 1 2>/dev/null | tail -1 | sed -e 's/^ *//' | sed -e 's/, "us_per_layer.*//'; }
echo "# qkv o gate_up down"
run 0 8 0 0
for o in 4 16; do run 0 $o 0 0; done
for d in 8 14 16 28; do run 0 8 0 $d; done
for g in 4 8 16; do run 0 8 $g 0; done
for q in 4 8 16; do run $q 8 0 0; done
run 0 8 0 0
