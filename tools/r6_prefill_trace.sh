#!/bin/bash
# per-kernel time of one prefill chunk (4 layers, tools/prof_prefill.py) with the own GEMM route and with the library route: where the chunk's difference sits
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pftrace; mkdir -p $O $R/gpurun_out/r6
for E in ${ROUTES:-1 0}; do
  EXL3_HIP_GEMM_NT=$E timeout 300 rocprofv3 --kernel-trace --stats -d $O/e$E -o out --output-format csv -- python $R/tools/prof_prefill.py > $O/e$E.log 2>&1
  python - <<PY
import csv
rows = list(csv.DictReader(open("$O/e$E/out_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("EXL3_HIP_GEMM_NT=$E total kernel time (2 chunks x 4 layers) us:", round(tot / 1e3, 1))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print("   %-60s calls %4s  total_us %9.1f  avg_us %8.1f" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
done 2>&1 | tee $R/gpurun_out/r6/prefill_trace_own_vs_lib.txt
rm -rf $O
