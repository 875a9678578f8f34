#!/bin/bash
# round 5, call 11: count matrices written by the int8 product (nm_direct16) -- tests, then the categorical bench A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run11; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_categorical.py tests/test_gpu_hoc.py -x -q -m gpu > $O/tests.txt 2>&1; tail -15 $O/tests.txt
for v in 1 0; do
CAT_NM_DIRECT16=$v timeout 300 python tools/categorical_bench.py 2>&1 | tail -1 > $O/cat_1000_d$v.json
CAT_NM_DIRECT16=$v timeout 300 python tools/categorical_bench.py 5000 2>&1 | tail -1 > $O/cat_5000_d$v.json
done
cat $O/cat_*.json
(cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/cp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/cp -o cp -- python $GRAFT_REPO_ROOT/tools/categorical_bench.py > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/kernel_table.py /tmp/cp > $GRAFT_REPO_ROOT/$O/categorical_kernels.txt 2>&1)
head -12 $O/categorical_kernels.txt
