#!/bin/bash
# Whole-step A/B of kernel micro-variants with the C++ decode-step harness (no Python on the GPU box; ~2.5 s per run).
#   here (CPU box):   tools/experiments/variant_queue.sh build      -> build/lite_<tag>/libexl3_hip.so for every variant below (7 MB each, ~30 s each)
#                     + the harness binary under tools/bin/
#   on the GPU box:   gpurun -- 'tools/experiments/variant_queue.sh run > gpurun_out/variant_queue.txt'
# Each variant runs between two runs of the base library (same box, alternating); the table is tok/s of the three-launch-MLP step.
# Variants = compile-time macros of the shipped kernels, all OFF in the product build (profiles/NOTES.md B section 6):
VARIANTS=(
  "plain:-DEXL3_LOAD_PLAIN"          # weight rows with the default cache policy instead of non-temporal loads
  "nosched:-DG4_NO_SCHED"            # generation 4: no sched_barrier per decoded quad (free instruction scheduling)
  "sched2:-DG4_SCHED2"               # ... one per two quads (16 weights in flight)
  "wpe1:-DG4_WPE_DELTA=1"            # ... one wave per SIMD less in every instantiation (larger register budget)
  "prio:-DG4_PRIO"                   # ... s_setprio 3 around the preparation tasks and the output half-waves
)
cd "$(dirname "$0")/../.." || exit 1
H=tools/bin/decode_step_harness
case "$1" in
  build)
    mkdir -p tools/bin
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-gpu-rdc -DM1_TAGGED -I include -I exllamav3_amd/csrc -o $H \
        tools/experiments/decode_step_harness.hip tools/experiments/exl3_mlp1.hip -L exllamav3_amd -lexl3_hip -Wl,-rpath,'$ORIGIN/../../exllamav3_amd' || exit 1
    python tools/experiments/build_lite.py base || exit 1
    for v in "${VARIANTS[@]}"; do python tools/experiments/build_lite.py "${v%%:*}" ${v#*:} || exit 1; done
    ;;
  run)
    one() { LD_LIBRARY_PATH=build/lite_$1 $H 32 2 0 8 0 0 1 2>&1 | tail -1 | sed -e 's/.*"tok_s": {"three_launch_mlp": \([0-9.]*\).*/\1/'; }
    echo "# variant  base_before  variant  base_after   (tok/s, Llama-3.1-8B bs 1 fx step, three-launch MLP)"
    for v in "${VARIANTS[@]}"; do t=${v%%:*}; echo "$t $(one base) $(one $t) $(one base)"; done
    ;;
  *) echo "usage: $0 build | run"; exit 2 ;;
esac
