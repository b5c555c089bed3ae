#!/bin/bash
# memory-path counters of the prefill GEMMs on the o_proj shape: generation 2 (GEN=2) / generation 1 against the library's kernel -> gpurun_out/r6/pmc_gemm_mem.json
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcm; mkdir -p $O $R/gpurun_out/r6
cat > /tmp/gm.py <<PY
import sys, os, torch
sys.path.insert(0, "$R")
from exllamav3_amd import ext
dev = torch.device("cuda:0"); ext.init(0); ext._GEMM_NT_OWN = False
M, k, n = 4096, 4096, 4096
a = torch.randn((M, k), device=dev).half(); bt = (torch.randn((n, k), device=dev) * 0.02).half()
c = torch.empty((M, n), dtype=torch.half, device=dev)
for _ in range(3): ext.hgemm_nt(a, bt, c)
torch.cuda.synchronize()
for _ in range(4): ext.gemm_nt_mfma(a, bt, c, 0, 2)
for _ in range(4): ext.gemm_nt_mfma(a, bt, c, 0, 1)
for _ in range(4): ext.hgemm_nt(a, bt, c)
torch.cuda.synchronize()
PY
i=0
for C in "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum"; do i=$((i+1))
  timeout 200 rocprofv3 --pmc $C -d $O/p$i -o out --output-format csv -- python /tmp/gm.py > $O/p$i.log 2>&1 || tail -3 $O/p$i.log
done
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "exl3_gemm_nt" in n or "Custom_Cijk" in n:
            acc[n.split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: round(sum(v[1:]) / max(len(v) - 1, 1), 1) for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open("$R/gpurun_out/r6/pmc_gemm_mem.json", "w"), indent=1)
for k, d in out.items():
    print(k); print("  ", d)
PY
rm -rf $O
