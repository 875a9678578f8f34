#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_solver_wave16.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/solver_lv_sweep.py 2>/dev/null | cut -c1-260 | tail -5
PLSPM_HIP_LIB=plspm-python_amd/csrc/build/marks/libplspm_hip_marks.so python tools/experiments/solver_marks_lv.py 20 3 2>&1 | tail -4 | cut -c1-250
PLSPM_HIP_LIB=plspm-python_amd/csrc/build/marks/libplspm_hip_marks.so python tools/experiments/solver_marks_lv.py 16 4 2>&1 | tail -4 | cut -c1-250
