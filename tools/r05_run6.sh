#!/bin/bash
R="$GRAFT_REPO_ROOT"; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/r05_run6; rm -rf $O; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_gram_i8.py -m gpu -q -x -k "persistent or plane_count" 2>&1 | tail -30 > $O/pytest_persist.txt
PLSPM_HIP_LIB=$R/plspm-python_amd/csrc/build/exp_i8/libplspm_hip_exp.so timeout 300 python tools/persist_ab.py 5000 > $O/persist_ab_exp.jsonl 2>$O/persist_ab.err
PLSPM_HIP_LIB=$R/plspm-python_amd/csrc/build/exp_i8/libplspm_hip_exp.so timeout 300 python tools/persist_ab.py 2500 >> $O/persist_ab_exp.jsonl 2>>$O/persist_ab.err
PLSPM_HIP_LIB=$R/plspm-python_amd/csrc/build/exp_i8/libplspm_hip_exp.so timeout 300 python tools/persist_ab.py 5000 7 >> $O/persist_ab_exp.jsonl 2>>$O/persist_ab.err
tail -5 $O/pytest_persist.txt; cat $O/persist_ab_exp.jsonl
