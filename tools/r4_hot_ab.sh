#!/bin/bash
# Whole-step A/B (C++ harness, Llama-3.1-8B bs 1 fx step) of the speed-only ablations G4_ABL_HOT_FIRST / G4_ABL_HOT_ALL (generation 4's first ring of weight
# rows / every weight row read from an L2-resident location): the upper bound for "the predecessor's drain warms the successor's first rows".
cd "$(dirname "$0")/.." || exit 1
H=tools/bin/decode_step_harness
one() { LD_LIBRARY_PATH=build/lite_$1 $H 32 2 0 8 0 0 1 2>&1 | tail -1 | sed -e 's/.*"tok_s": {"three_launch_mlp": \([0-9.]*\).*/\1/'; }
echo "# variant  base_before  variant  base_after   (tok/s)"
for t in hotfirst hotall hotfirst hotall; do echo "$t $(one base) $(one $t) $(one base)"; done
