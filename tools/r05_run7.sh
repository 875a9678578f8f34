#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/r05_run7; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_categorical.py tests/test_gpu_hoc.py -m gpu -q 2>&1 | tail -60 > $O/pytest_cat.txt
timeout 300 python tools/categorical_bench.py > $O/cat_bench.json 2>$O/cat_bench.err
timeout 300 python tools/categorical_bench.py 5000 >> $O/cat_bench.json 2>>$O/cat_bench.err
timeout 300 python tools/hoc_bench.py 2>&1 | tail -2 > $O/hoc_bench.json
tail -15 $O/pytest_cat.txt; cat $O/cat_bench.json; cat $O/hoc_bench.json
