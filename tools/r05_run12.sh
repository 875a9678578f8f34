#!/bin/bash
# round 5, call 12: full GPU suite + HOC bench + kernel table of the HOC-on-ORD bootstrap
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run12; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -5 $O/tests.txt
timeout 600 python tools/hoc_bench.py 2>&1 | tail -1 > $O/hoc_bench.json; cat $O/hoc_bench.json
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/cp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/cp -o cp -- python $GRAFT_REPO_ROOT/tools/hoc_bench.py > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/kernel_table.py /tmp/cp > $GRAFT_REPO_ROOT/$O/hoc_kernels.txt 2>&1)
head -24 $O/hoc_kernels.txt
