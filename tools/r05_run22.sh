#!/bin/bash
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_run22; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/cp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/cp -o cp -- python $R/tools/hoc_bench.py > /dev/null 2>&1; python $R/tools/kernel_table.py /tmp/cp > $O/hoc_kernels.txt 2>&1)
head -24 $O/hoc_kernels.txt
