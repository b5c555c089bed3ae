#!/bin/bash
# Round-3 profile refresh + PMC accounting: gpurun -- 'bash tools/r3_profiles.sh'
cd $GRAFT_REPO_ROOT
bash tools/final_profiles.sh r03 2>&1 | tail -25
bash tools/pmc_decode.sh 2>&1 | tail -12
EXL3_HIP_GEMV_GEN4=0 bash -c 'sed "s#gpurun_out/pmc_decode.json#gpurun_out/pmc_decode_gen2.json#; s#gpurun_out/pmcd#gpurun_out/pmcd2#" tools/pmc_decode.sh > /tmp/pmc2.sh; sed -i "s#bench.py --no-extra#bench.py --pipeline glue --no-extra#" /tmp/pmc2.sh; bash /tmp/pmc2.sh' 2>&1 | tail -8
