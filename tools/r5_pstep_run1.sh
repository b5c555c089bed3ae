#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5; mkdir -p $O
export LD_LIBRARY_PATH=$PWD/exllamav3_amd:$LD_LIBRARY_PATH
H=tools/bin/pstep_harness
timeout 120 $H 8b 1 1 "0,1,2" > $O/b_8b_l1.json 2> $O/b_8b_l1.err; echo "rc=$?" >> $O/b_8b_l1.err
timeout 120 $H 8b 2 1 "0,1,2" > $O/b_8b_l2.json 2> $O/b_8b_l2.err; echo "rc=$?" >> $O/b_8b_l2.err
timeout 120 $H 1b 2 1 "0,1,2" > $O/b_1b_l2.json 2> $O/b_1b_l2.err; echo "rc=$?" >> $O/b_1b_l2.err
timeout 240 $H 8b 0 3 "2,1,0" $O/ps_8b_stamps.bin > $O/b_8b.json 2> $O/b_8b.err; echo "rc=$?" >> $O/b_8b.err
timeout 240 $H 1b 0 3 "2,1,0" $O/ps_1b_stamps.bin > $O/b_1b.json 2> $O/b_1b.err; echo "rc=$?" >> $O/b_1b.err
tail -c 1500 $O/b_*_l?.json $O/b_*_l?.err
tail -c 2500 $O/b_8b.json $O/b_8b.err $O/b_1b.json $O/b_1b.err
