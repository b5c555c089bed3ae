#!/bin/bash
# round 4: in-kernel ablations of the generation-3 streaming loop at batch 16 (lite libraries built with -DG3_ABL_*): per-launch kernel durations from rocprofv3
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4abl; rm -rf $O; mkdir -p $O
for v in g3w ablhot ablnomfma ablnolds ablnodec ablnodechot; do
  LD_LIBRARY_PATH=$R/build/lite_$v EXL3_HIP_GEMM3_CPW=1 timeout 120 rocprofv3 --kernel-trace -d $O/$v -o out --output-format csv -- $R/tools/bin/decode_step_harness 8 1 0 0 0 0 1 16 > $O/$v.log 2>&1
  python3 - <<PY
import csv, glob, collections
dur = collections.defaultdict(list)
for f in glob.glob("$O/$v/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "gemm3" in n:
            gs = r.get("Grid_Size") or str(int(r.get("Grid_Size_X", 0)) * int(r.get("Grid_Size_Y", 1)))
            dur[gs].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
names = {"98304": "qkv", "90112": "o", "114688": "gate_up", "131072": "down", "256512": "lm_head"}
print("$v", {names.get(k, k): round(sorted(v)[len(v) // 2], 2) for k, v in sorted(dur.items())})
PY
done | tee $R/gpurun_out/r4_g3_ablate.txt
rm -rf $O
