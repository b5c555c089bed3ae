#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3m; mkdir -p $O; cd $R
for abl in 0 1 2 4 8 12 15 3; do echo -n "abl=$abl "; EXL3_HIP_LIB=$R/build/libexl3_hip_abl.so EXL3_HIP_RECON_ABL=$abl timeout 120 python tools/bench_recon.py 2>/dev/null | sed -n 3p; done | tee $O/abl2.txt
