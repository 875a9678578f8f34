#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/r05_run3; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -150 > $O/pytest.txt
tail -12 $O/pytest.txt
