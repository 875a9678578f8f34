#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run16; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/cp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/cp -o cp -- python $GRAFT_REPO_ROOT/tools/hoc_bench.py > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/kernel_table.py /tmp/cp > $GRAFT_REPO_ROOT/$O/hoc_kernels.txt 2>&1)
head -24 $O/hoc_kernels.txt
