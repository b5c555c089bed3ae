# PMC passes + kernel trace of the generation-3 small-m GEMM on the gate|up shape at 16 rows, and kernel stats of the bs-16 decode bench.
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof7; mkdir -p $O
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
C2="SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES"
C3="FETCH_SIZE WRITE_SIZE"
A="--n 28672 --m 16 --iters 6"
rocprofv3 --pmc $C1 -d $O/p1 -o out --output-format csv -- python $R/tools/prof_one.py $A > /dev/null 2>&1
rocprofv3 --pmc $C2 -d $O/p2 -o out --output-format csv -- python $R/tools/prof_one.py $A > /dev/null 2>&1
rocprofv3 --pmc $C3 -d $O/p3 -o out --output-format csv -- python $R/tools/prof_one.py $A > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt -o out --output-format csv -- python $R/tools/prof_one.py $A > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/bench16 -o out --output-format csv -- python $R/bench.py --batch 16 --no-prefill --no-cpu --steps 20 > $O/bench16.log 2>&1
ls $O $O/*
