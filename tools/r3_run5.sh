#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest.txt 2>&1; echo "tests rc=$?" | tee -a $O/pytest.txt
tail -12 $O/pytest.txt | cut -c1-300
REPS=3 bash tools/r3_ab_pipelines.sh "fx6:A=1:--pipeline fx" "fx5_act:EXL3_HIP_FX_ACT_IN_GEMV=1:--pipeline fx"
