#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/r05_run4; rm -rf $O; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_gram_i8.py -m gpu -q -x -k "persistent" 2>&1 | tail -40 > $O/pytest_persist.txt
timeout 300 python tools/persist_ab.py 5000 > $O/persist_ab.jsonl 2>$O/persist_ab.err
timeout 300 python tools/persist_ab.py 5000 7 >> $O/persist_ab.jsonl 2>>$O/persist_ab.err
timeout 300 python tools/persist_ab.py 2500 >> $O/persist_ab.jsonl 2>>$O/persist_ab.err
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/pytest.txt
tail -5 $O/pytest_persist.txt; cat $O/persist_ab.jsonl; tail -8 $O/pytest.txt
