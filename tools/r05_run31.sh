#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_solver_wave16.py tests/test_gpu_solver_wave.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8
timeout 600 python tools/solver_lv_sweep.py 2>/dev/null | cut -c1-260 | tail -3
