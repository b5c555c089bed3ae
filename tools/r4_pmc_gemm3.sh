#!/bin/bash
# round 4: PMC passes over the batch-16 decode step (C++ harness, 4 layers, eager step + graph replays), counters averaged per (kernel, grid) --
# the generation-3 launches q|k|v / o / gate|up / down are told apart by their grid.  Separate --pmc passes, no trace domains (gpurun rule).
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4pmc; rm -rf $O; mkdir -p $O
H="$R/tools/bin/decode_step_harness 4 1 0 0 0 0 1 16"
export LD_LIBRARY_PATH=$R/build/lite_${1:-g3late}
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM"
P3="FETCH_SIZE GRBM_GUI_ACTIVE"
P4="TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
P5="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES"
i=0
for C in "$P1" "$P2" "$P3" "$P4" "$P5"; do i=$((i+1))
  timeout 120 rocprofv3 --pmc $C -d $O/p$i -o out --output-format csv -- $H > $O/p$i.log 2>&1 || echo "pass $i failed" >> $O/fail.log
done
timeout 120 rocprofv3 --kernel-trace --stats -d $O/kt -o out --output-format csv -- $H > $O/kt.log 2>&1
python3 - <<PY
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "exl3_" in n or "glue_" in n:
            key = n.split("(")[0][:70] + " grid=" + r.get("Grid_Size", "?") + " wg=" + r.get("Workgroup_Size", "?")
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v), 1) for c, v in d.items()} | {"dispatches": max(len(v) for v in d.values())} for k, d in acc.items()}
dur = collections.defaultdict(list)
for f in glob.glob("$O/kt/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "exl3_" in n or "glue_" in n:
            gs = r.get("Grid_Size") or str(int(r.get("Grid_Size_X", 0)) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1)))
            key = n.split("(")[0][:70] + " grid=" + gs + " wg=" + (r.get("Workgroup_Size") or r.get("Workgroup_Size_X", "?"))
            dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
for k, v in dur.items():
    v.sort(); out.setdefault(k, {})["kernel_us_median"] = round(v[len(v) // 2], 2); out[k]["kernel_us_n"] = len(v)
json.dump(out, open("$R/gpurun_out/r4pmc.json", "w"), indent=1)
for k, d in sorted(out.items()):
    print(k, {c: d.get(c) for c in ("kernel_us_median", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "FETCH_SIZE", "GRBM_GUI_ACTIVE")})
PY
cp $O/kt/*kernel_stats.csv $R/gpurun_out/r4pmc_kernel_stats.csv 2>/dev/null
ls $O/kt | head
