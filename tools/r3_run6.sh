#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3i; mkdir -p $O; cd $R
timeout 600 python -m pytest tests -m gpu -q --maxfail=5 -k "fixed_point or gemv4 or resid_in_gemv" 2>&1 | tail -12 | cut -c1-400
REPS=3 bash tools/r3_ab_pipelines.sh "fx6:A=1:--pipeline fx" "fx5_gu_atomic:EXL3_HIP_FX_GU_ATOMIC=1:--pipeline fx"
