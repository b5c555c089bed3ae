#!/bin/bash
# generation-3 ablation libraries on the bs-16 decode step (tools/sweep_split.py: whole-step time per layer under graph replay)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r3s
for rep in 1 2; do for L in "" g3a g3b g3c; do
  LIB=$R/exllamav3_amd/libexl3_hip.so; [ -n "$L" ] && LIB=$R/build/libexl3_hip_$L.so
  echo "== ${L:-base}"; EXL3_HIP_LIB=$LIB BSZ=16 CANDS='{"gu":[2,4,8],"qkv":[16],"down":[28]}' timeout 100 python tools/sweep_split.py 16 2>&1 | grep -v amdgpu.ids
done; done | tee gpurun_out/r3s/g3ab.txt
