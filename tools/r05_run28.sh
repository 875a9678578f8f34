#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run28; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_solver_wave16.py tests/test_gpu_solver_wave.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
(AB_MODES=BBBBBB python tools/aux_ab.py solver_wave=0,3,1 2>&1 | grep "^{"; AB_MODES=ABABAB python tools/aux_ab.py solver_wave=0,3,1 2>&1 | grep "^{") > $O/ab_solver_wave_mode_b.jsonl; cut -c1-200 $O/ab_solver_wave_mode_b.jsonl
PLSPM_HIP_LIB=plspm-python_amd/csrc/build/marks/libplspm_hip_marks.so python tools/wave_marks.py 5000 2>&1 | grep -A4 "modes BBBBBB" | tail -4 | cut -c1-300
