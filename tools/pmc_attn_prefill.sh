# PMC passes over exl3_attn_prefill_paged alone (tools/bench_attn_prefill.py): counters averaged per dispatch.
# gpurun -- 'bash tools/pmc_attn_prefill.sh'  -> gpurun_out/pmc_attn_prefill.json
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmca; mkdir -p $O
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
C2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"
C3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES"
i=0
for C in "$C1" "$C2" "$C3"; do i=$((i+1))
  timeout 200 rocprofv3 --pmc $C -d $O/p$i -o out --output-format csv -- python $R/tools/bench_attn_prefill.py 4096 0 > $O/p$i.log 2>&1
done
python - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "attn_prefill" in n:
            acc[n.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v), 1) for c, v in d.items()} | {"dispatches": max(len(v) for v in d.values())} for k, d in acc.items()}
json.dump(out, open("$R/gpurun_out/pmc_attn_prefill.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
tail -3 $O/p3.log
rm -rf $O
