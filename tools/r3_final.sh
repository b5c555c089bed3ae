#!/bin/bash
# round-end validation: the whole GPU suite, smoke, and the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final2; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.err; python - <<PY
import json
d = json.loads([l for l in open("$O/bench_default.json") if l.startswith("{")][-1])
print("decode", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "prefill", d["prefill"]["value"], d["prefill"]["roofline"]["frac"])
print({k: v.get("tok_s") for k, v in d["other_configs"].items()})
PY
