#!/bin/bash
# shader clock under the prefill GEMMs: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / kernel duration of the same dispatch, generation 2 against the library (o_proj shape)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcclk; mkdir -p $O $R/gpurun_out/r6
cat > /tmp/gc.py <<PY
import sys, os, torch
sys.path.insert(0, "$R")
from exllamav3_amd import ext
dev = torch.device("cuda:0"); ext.init(0); ext._GEMM_NT_OWN = False
M, k, n = 4096, 4096, 4096
a = torch.randn((M, k), device=dev).half(); bt = (torch.randn((n, k), device=dev) * 0.02).half()
c = torch.empty((M, n), dtype=torch.half, device=dev)
for _ in range(3): ext.hgemm_nt(a, bt, c)
torch.cuda.synchronize()
for r in range(3):
    for _ in range(10): ext.gemm_nt_mfma(a, bt, c, 0, 2)
    for _ in range(10): ext.hgemm_nt(a, bt, c)
torch.cuda.synchronize()
PY
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/p -o out --output-format csv -- python /tmp/gc.py > $O/p.log 2>&1
python - <<PY
import csv, glob, collections, statistics
dur = {}
for f in glob.glob("$O/p/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(list)
for f in glob.glob("$O/p/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and r["Dispatch_Id"] in dur:
            n, d = dur[r["Dispatch_Id"]]
            if "exl3_gemm_nt" in n or "Custom_Cijk" in n:
                acc[n.split("(")[0][:50]].append((float(r["Counter_Value"]), d))
for k, v in acc.items():
    v = v[len(v) // 2:]
    print(k, "n", len(v), "cycles/8", round(statistics.median(x[0] for x in v) / 8), "dur_us", round(statistics.median(x[1] for x in v) / 1e3, 1), "MHz", round(statistics.median(x[0] / 8 / x[1] * 1e3 for x in v)))
PY
tail -2 $O/p.log; rm -rf $O
