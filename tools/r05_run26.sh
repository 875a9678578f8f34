#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run26; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_solver_wave16.py tests/test_gpu_solver_wave.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -m gpu > $O/tests.txt 2>&1; tail -8 $O/tests.txt
AB_MODES=BBBBBB python tools/aux_ab.py solver_wave=0,3,1 2>&1 | grep "^{" | cut -c1-330
AB_MODES=ABABAB python tools/aux_ab.py solver_wave=3,1 2>&1 | grep "^{" | cut -c1-330
