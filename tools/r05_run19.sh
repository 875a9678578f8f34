#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run19; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_solver_quad.py -x -q -m gpu > $O/tests.txt 2>&1; tail -15 $O/tests.txt
timeout 600 python tools/solver_quad_ab.py > $O/quad_ab.jsonl 2> $O/quad_ab.err; cat $O/quad_ab.jsonl; tail -3 $O/quad_ab.err
PLSPM_HIP_LIB=plspm-python_amd/csrc/build/marks/libplspm_hip_marks.so timeout 300 python tools/experiments/solver_marks_120.py > $O/marks.txt 2>&1; tail -12 $O/marks.txt
