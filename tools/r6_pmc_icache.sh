#!/bin/bash
# SQ / instruction-cache counters of the persistent step kernel (C++ harness, persistent side only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r6; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/${LIBDIR:-exllamav3_amd}:$LD_LIBRARY_PATH
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc1 /tmp/pmc2
H_ONLY=ps H_SPIN_LIMIT=200000 timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace -d /tmp/pmc1 -o p1 --output-format csv -- $GRAFT_REPO_ROOT/tools/bin/pstep_harness ${1:-8b} 0 1 "3" > $O/pmc1.log 2>&1
H_ONLY=ps H_SPIN_LIMIT=200000 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace -d /tmp/pmc2 -o p2 --output-format csv -- $GRAFT_REPO_ROOT/tools/bin/pstep_harness ${1:-8b} 0 1 "3" > $O/pmc2.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("/tmp/pmc1", "/tmp/pmc2"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "pstep_kernel" not in r.get("Kernel_Name", ""): continue
            acc[r["Counter_Name"]][0] += float(r["Counter_Value"]); acc[r["Counter_Name"]][1] += 1
        for k, (v, n) in sorted(acc.items()): print(d, k, "per launch", v / max(n, 1), "launches", n)
PY
