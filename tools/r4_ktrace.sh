#!/bin/bash
# round 4: per-kernel durations (rocprofv3 --kernel-trace) of the batch-16 decode step in the C++ harness; usage: r4_ktrace.sh LIBTAG "ENV=.. ENV=.." [harness args...]
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4kt_$$; rm -rf $O; mkdir -p $O
LIB=$1; shift; ENVS=$1; shift
env LD_LIBRARY_PATH=$R/build/lite_$LIB $ENVS timeout 120 rocprofv3 --kernel-trace -d $O -o out --output-format csv -- $R/tools/bin/decode_step_harness ${@:-8 1 0 0 0 0 1 16} > $O/run.log 2>&1
python3 - <<PY
import csv, glob, collections
dur = collections.defaultdict(list)
for f in glob.glob("$O/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        if "exl3_" in n or "glue_" in n or "fx_" in n:
            gs = int(r.get("Grid_Size") or (int(r.get("Grid_Size_X", 0)) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))))
            wg = int(r.get("Workgroup_Size") or (int(r.get("Workgroup_Size_X", 1)) * int(r.get("Workgroup_Size_Y", 1))))
            dur[(n.split("(")[0].replace("void ", "")[:44], gs // max(wg, 1), wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
tot = 0.0
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v.sort(); print("   %-46s wgs=%-5d thr=%-5d n=%-4d median %.2f us  sum %.1f" % (k[0], k[1], k[2], len(v), v[len(v) // 2], sum(v))); tot += sum(v)
print("   total kernel time", round(tot, 1))
PY
tail -1 $O/run.log | cut -c1-200
rm -rf $O
