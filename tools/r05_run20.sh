#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run20; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_solver_quad.py tests/test_gpu_fuzz.py tests/test_gpu_gram_i8.py tests/test_gpu_solver_wave.py -x -q -m gpu > $O/tests.txt 2>&1; tail -5 $O/tests.txt
timeout 600 python tools/solver_quad_ab.py > $O/quad_ab.jsonl 2> $O/quad_ab.err; cut -c1-220 $O/quad_ab.jsonl; tail -3 $O/quad_ab.err
PLSPM_HIP_LIB=plspm-python_amd/csrc/build/marks/libplspm_hip_marks.so timeout 300 python tools/experiments/solver_marks_120.py > $O/marks.txt 2>&1; tail -4 $O/marks.txt
