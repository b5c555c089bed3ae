#!/bin/bash
# PMC passes over the hand-written NT GEMM and the library route on the o_proj shape (4096^3): gpurun -- 'bash tools/pmc_gemm_nt.sh' -> gpurun_out/r6/pmc_gemm_nt2.json
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcg; mkdir -p $O
cat > /tmp/g.py <<PY
import sys, os, torch
sys.path.insert(0, "$R")
from exllamav3_amd import ext
dev = torch.device("cuda:0"); ext.init(0); ext._GEMM_NT_OWN = False
M, k, n = 4096, 4096, 4096
a = torch.randn((M, k), device=dev).half(); bt = (torch.randn((n, k), device=dev) * 0.02).half()
c = torch.empty((M, n), dtype=torch.half, device=dev)
for _ in range(3): ext.gemm_nt_mfma(a, bt, c, 0, int(os.environ.get("GEN", "0")))
for _ in range(3): ext.hgemm_nt(a, bt, c)
torch.cuda.synchronize()
PY
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
C2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU"
C3="SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
i=0
for C in "$C1" "$C2" "$C3"; do i=$((i+1))
  timeout 200 rocprofv3 --pmc $C -d $O/p$i -o out --output-format csv -- python /tmp/g.py > $O/p$i.log 2>&1
done
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "exl3_gemm_nt" in n or "Cijk" in n:
            acc[n.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v), 1) for c, v in d.items()} for k, d in acc.items()}
json.dump(out, open("$R/gpurun_out/r6/pmc_gemm_nt2.json", "w"), indent=1)
for k, d in out.items():
    print(k); print("  ", d)
PY
tail -3 $O/p1.log
rm -rf $O
