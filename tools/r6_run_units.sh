cd $GRAFT_REPO_ROOT; O=gpurun_out/r6; mkdir -p $O
for T in units unitshot; do
LD_LIBRARY_PATH=$PWD/build/alt_$T:$LD_LIBRARY_PATH H_SPIN_LIMIT=20000 timeout 90 tools/bin/pstep_harness 8b 0 1 "3" $O/s_$T.bin 2>&1 | grep -o '"best".*' | cut -c1-200
echo "== $T"; python3 tools/pstep_unit_timeline.py $O/s_$T.bin 32 34; rm -f $O/s_$T.bin
done
