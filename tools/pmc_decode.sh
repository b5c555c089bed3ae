# PMC passes over the bs-1 decode step (eager launches: one record per dispatch), counters averaged per gen-2 GEMV instantiation.
# gpurun -- 'bash tools/pmc_decode.sh'  -> gpurun_out/pmc_decode.json
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcd; mkdir -p $O
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
C2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"
C3="FETCH_SIZE WRITE_SIZE"
i=0
for C in "$C1" "$C2" "$C3"; do i=$((i+1))
  timeout 300 rocprofv3 --pmc $C -d $O/p$i -o out --output-format csv -- python $R/bench.py --no-extra --no-cpu --no-prefill --steps 3 --warmup 1 --no-graph > $O/p$i.log 2>&1
done
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "exl3_gemv2_kernel" in n or "exl3_gemv4_kernel" in n or "glue_" in n:
            acc[n.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v), 1) for c, v in d.items()} | {"dispatches": max(len(v) for v in d.values())} for k, d in acc.items()}
json.dump(out, open("$R/gpurun_out/pmc_decode.json", "w"), indent=1)
for k, d in out.items(): print(k[:60], d.get("SQ_INSTS_VALU"), d.get("SQ_INSTS_MFMA"), d.get("SQ_ACTIVE_INST_VALU"), d.get("SQ_BUSY_CYCLES"), d.get("FETCH_SIZE"))
PY
rm -rf $O
