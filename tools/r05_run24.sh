#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run24; mkdir -p $O
python tools/aux_ab.py solver_wave=1,2 2>&1 | tail -2 | cut -c1-400
python tools/aux_ab.py solver_wave=2,1 2>&1 | tail -2 | cut -c1-400
PLSPM_HIP_LIB=plspm-python_amd/csrc/build/marks/libplspm_hip_marks.so python tools/experiments/solver_marks.py solver_wave=2 2>&1 | grep -v summary | tail -3
timeout 600 python tools/solver_lv_sweep.py 2>/dev/null | cut -c1-260
