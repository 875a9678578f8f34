#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run30; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH -d $O/p1 -o p1 -- python $R/tools/size_rows.py > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $O/p2 -o p2 -- python $R/tools/size_rows.py > /dev/null 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $O/p3 -o p3 -- python $R/tools/size_rows.py > /dev/null 2>&1
cd $R; python tools/pmc_rows.py $O solver_quad solver_wave16 > $O/solver_sizes_pmc.txt 2>&1; find $O -name "*.db" -delete; cat $O/solver_sizes_pmc.txt | head -70
