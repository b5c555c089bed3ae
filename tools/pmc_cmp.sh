cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof6; mkdir -p $O
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
C2="SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC"
rocprofv3 --pmc $C1 -d $O/k1 -o out --output-format csv -- python $R/tools/prof_one.py --iters 6 > /dev/null 2>&1
rocprofv3 --pmc $C2 -d $O/k2 -o out --output-format csv -- python $R/tools/prof_one.py --iters 6 > /dev/null 2>&1
rocprofv3 --pmc $C1 -d $O/m1 -o out --output-format csv -- $R/tools/bin/ubench_mix > /dev/null 2>&1
rocprofv3 --pmc $C2 -d $O/m2 -o out --output-format csv -- $R/tools/bin/ubench_mix > /dev/null 2>&1
ls $O
