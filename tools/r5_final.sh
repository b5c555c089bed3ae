#!/bin/bash
# Round-5 round-end evidence (run on the GPU box: gpurun -- 'bash tools/r5_final.sh'): the GPU test suite, rocprofv3 kernel stats of the bench configs (each under its
# own timeout), a separate PMC pass (FETCH_SIZE only, no tracing) for roofline.traffic, the default bench line.  Results: gpurun_out/final/ -> copy to profiles/.
TAG=r05
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O
cd $R
if [ "${1:-all}" != "noprof" ]; then
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > $O/${TAG}_gpu_tests.txt; cat $O/${TAG}_gpu_tests.txt
fi
cd /tmp && export TMPDIR=/tmp
run() { # name, bench args...
  n=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/$n -o out --output-format csv -- python $R/bench.py --no-extra --no-cpu "$@" > $O/${TAG}_bench_${n}_under_rocprof.json 2> $O/$n.err
  cp $O/$n/out_kernel_stats.csv $O/${TAG}_bench_${n}_kernel_stats.csv 2>/dev/null
  rm -rf $O/$n
}
run bs1 --no-prefill --steps 20
run bs1_launch_per_op --pipeline fx --no-prefill --steps 20
run 1b_persistent --model llama-3.2-1b --no-prefill --steps 20
run 1b_launch_per_op --model llama-3.2-1b --pipeline fx --no-prefill --steps 20
run bs1_attention --attention --no-prefill --steps 20
run bs1_attention_launch_per_op --attention --pipeline fx --no-prefill --steps 20
run prefill --steps 5
# PMC passes: eager launches (one dispatch record per kernel), FETCH_SIZE in KiB, x2 on gfx950 for wide coalesced reads (MI355X_MICROARCH.md).  One pass per pipeline:
# the launch-per-op GEMV launches, the persistent step's ONE kernel (8B and 1B)
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc -o out --output-format csv -- python $R/bench.py --no-extra --no-cpu --no-prefill --steps 3 --warmup 1 --no-graph --pipeline fx > $O/pmc.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_p8 -o out --output-format csv -- python $R/bench.py --no-extra --no-cpu --no-prefill --steps 3 --warmup 1 --no-graph > $O/pmc_p8.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_p1 -o out --output-format csv -- python $R/bench.py --no-extra --no-cpu --no-prefill --steps 3 --warmup 1 --no-graph --model llama-3.2-1b > $O/pmc_p1.log 2>&1
python - <<PY
import csv, glob, json
def rows_of(d):
    rows = []
    for f in glob.glob(d + "/*counter_collection.csv"):
        rows += list(csv.DictReader(open(f)))
    return rows
rows = rows_of("$O/pmc")
g = [r for r in rows if ("exl3_gemv4_kernel" in r.get("Kernel_Name", "") or "exl3_gemv2_kernel" in r.get("Kernel_Name", "")) and r.get("Counter_Name") == "FETCH_SIZE"]
out = {}
if g:
    per = sum(float(r["Counter_Value"]) for r in g) / len(g) * 1024 * 2
    ko = None
    try:
        kr = [r for r in csv.DictReader(open("$O/${TAG}_bench_bs1_launch_per_op_kernel_stats.csv")) if "exl3_gemv4_kernel" in r["Name"] or "exl3_gemv2_kernel" in r["Name"]]
        calls = sum(int(r["Calls"]) for r in kr); tot = sum(float(r["TotalDurationNs"]) for r in kr)
        ko = {"avg_launch_us": round(tot / calls / 1e3, 3), "calls": calls, "source": "profiles/${TAG}_bench_bs1_launch_per_op_kernel_stats.csv"}
    except Exception as e:
        print("kernel-only figure unavailable:", e)
    out = {"kernel": "exl3_gemv4_kernel<4,2,1,*> (all GEMV launches of the decode step, Llama-3.1-8B 4bpw mul1, bs=1, fx pipeline)", "fetch_bytes_per_launch": int(per),
           "launches_sampled": len(g), "collected": "${TAG}, tools/r5_final.sh", "kernel_only": ko, "method": "rocprofv3 --pmc FETCH_SIZE (own pass, no tracing) on bench.py --steps 3 --warmup 1 --no-graph; FETCH_SIZE is KiB and reads 1/2 of a wide coalesced stream on gfx950 (MI355X_MICROARCH.md, HBM section): x1024 x2 applied; average over all GEMV dispatches (includes the activation / residual-accumulator reads and scale vectors)"}
    print("traffic bytes/launch", int(per), "over", len(g))
else:
    print("no PMC rows (fx)", len(rows))
ps = {}
for model, d in (("llama-3.1-8b", "$O/pmc_p8"), ("llama-3.2-1b", "$O/pmc_p1")):
    r_ = [r for r in rows_of(d) if "exl3_pstep_kernel" in r.get("Kernel_Name", "") and r.get("Counter_Name") == "FETCH_SIZE"]
    if r_:
        per = sum(float(r["Counter_Value"]) for r in r_) / len(r_) * 1024 * 2
        ps[model] = {"fetch_bytes_per_launch": int(per), "launches_sampled": len(r_), "collected": "${TAG}, tools/r5_final.sh",
                     "method": "rocprofv3 --pmc FETCH_SIZE (own pass) on bench.py --no-graph --steps 3 --warmup 1; x1024 x2 as above; one dispatch = one decode step"}
        print("persistent step", model, "fetch bytes / launch", int(per), "over", len(r_))
    else:
        print("no PMC rows for the persistent step", model)
out["persistent_step"] = ps
json.dump(out, open("$O/traffic.json", "w"), indent=1)
PY
rm -rf $O/pmc $O/pmc_p8 $O/pmc_p1
cd $R; timeout 400 python bench.py > $O/${TAG}_bench_final.json 2> $O/final.err; tail -c 300 $O/final.err
ls $O; for f in $O/${TAG}_bench_*_under_rocprof.json $O/${TAG}_bench_final.json; do echo $f; grep "^{" $f | cut -c1-260; done
