#!/bin/bash
# quick categorical A/B: bench at 1,000 and 5,000 replicates per step + kernel table of the 1,000-replicate run
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_cat_quick; mkdir -p $O
timeout 300 python tools/categorical_bench.py 2>&1 | tail -1 > $O/cat_1000.json
timeout 300 python tools/categorical_bench.py 5000 2>&1 | tail -1 > $O/cat_5000.json
cat $O/cat_*.json
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/cp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/cp -o cp -- python $GRAFT_REPO_ROOT/tools/categorical_bench.py > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/kernel_table.py /tmp/cp > $GRAFT_REPO_ROOT/$O/categorical_kernels.txt 2>&1)
head -12 $O/categorical_kernels.txt
