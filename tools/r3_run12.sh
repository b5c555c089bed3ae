#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3l; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_norm.py -q -x > $O/t1.log 2>&1; tail -3 $O/t1.log
timeout 900 python -m pytest tests/test_gpu_path.py -q -x -k "bc_gated or mixtral or graph" > $O/t2.log 2>&1; tail -3 $O/t2.log
