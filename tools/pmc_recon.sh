#!/bin/bash
# SQ / LDS / memory counters of reconstruct_had_slice_t on the gate shape (4096 x 14336, K = 4, mul1): gpurun -- 'bash tools/pmc_recon.sh'
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcr; mkdir -p $O $R/gpurun_out/r6
cat > /tmp/rc.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from exllamav3_amd import ext
dev = torch.device("cuda:0"); ext.init(0)
k, n, K = 4096, 14336, 4
g = torch.Generator(device=dev); g.manual_seed(0)
trs = [torch.randint(-32768, 32768, (k // 16, n // 16, 16 * K), dtype=torch.int16, device=dev, generator=g) for _ in range(3)]
suh = torch.ones(k, device=dev).half(); svh = torch.ones(n, device=dev).half()
ws = [torch.empty((n, k), dtype=torch.half, device=dev) for _ in range(3)]
for i in range(6): ext.reconstruct_had_slice_t(ws[i % 3], trs[i % 3], suh, svh, K, False, True, 0)
torch.cuda.synchronize()
PY
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_LEVEL_WAVES SQ_ACCUM_PREV_HIRES" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do i=$((i+1))
  timeout 120 rocprofv3 --pmc $C -d $O/p$i -o out --output-format csv -- python /tmp/rc.py > $O/p$i.log 2>&1 || tail -2 $O/p$i.log
done
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "reconstruct_had_kernel" in r.get("Kernel_Name", ""): acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {c: round(sum(v[1:]) / max(len(v) - 1, 1), 1) for c, v in acc.items()}
json.dump(out, open("$R/gpurun_out/r6/pmc_recon.json", "w"), indent=1)
print(out)
PY
rm -rf $O
