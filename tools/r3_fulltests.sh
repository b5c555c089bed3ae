#!/bin/bash
# the whole GPU suite + smoke, as the driver runs them
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/full; mkdir -p $O; cd $R
timeout 2400 python -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -5 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
