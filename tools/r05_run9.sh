#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/r05_run9; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/pytest.txt
timeout 300 python tools/categorical_bench.py > $O/cat_bench.json 2>$O/cat_bench.err
timeout 300 python tools/categorical_bench.py 5000 >> $O/cat_bench.json 2>>$O/cat_bench.err
tail -6 $O/pytest.txt; cat $O/cat_bench.json
