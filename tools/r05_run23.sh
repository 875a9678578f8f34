#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run23; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_solver_wave16.py tests/test_gpu_solver_wave.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/tests.txt 2>&1; tail -15 $O/tests.txt
timeout 600 python tools/solver_lv_sweep.py > $O/lv_sweep.jsonl 2> $O/lv_sweep.err; cut -c1-260 $O/lv_sweep.jsonl; tail -3 $O/lv_sweep.err
