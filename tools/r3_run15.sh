#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3o; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/p -o out --output-format csv -- python $R/bench.py --batch 16 --layers 4 --no-prefill --no-extra --no-cpu --steps 20 > $O/b.json 2> $O/b.err
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$O/p/out_kernel_trace.csv")))
rows = [r for r in rows if "at::native" not in r["Kernel_Name"] and "rocclr" not in r["Kernel_Name"]]
# last 20 steps' worth: take the tail and print the sequence of one layer with durations
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "glue_qkv" in r["Kernel_Name"]]
i0 = idx[len(idx) // 2]
tail = rows[i0 - 2: i0 + 10]
for r in tail:
    print(r["Kernel_Name"][:60].ljust(60), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0, r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Workgroup_Size_X", ""))
PY
rm -rf $O/p
