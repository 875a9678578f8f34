#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/r05_run8; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/cp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/cp -o cp -- python $R/tools/categorical_bench.py > /dev/null 2>&1; python $R/tools/kernel_table.py /tmp/cp > $O/categorical_kernels.txt 2>&1
cat $O/categorical_kernels.txt | head -30
