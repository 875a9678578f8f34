#!/bin/bash
# solver work: phase clocks (marks build), parity tests that exercise the rows solver, interleaved A/B bench
O=gpurun_out/q; mkdir -p $O
PLSPM_HIP_LIB=plspm-python_amd/csrc/build/marks/libplspm_hip_marks.so timeout 300 python tools/experiments/solver_marks.py > $O/marks.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gram_i8.py tests/test_gpu_fuzz.py -q -x 2>&1 | tail -5 > $O/pytest_sel.txt
timeout 300 python tools/i8_bench.py 5000 3 2>&1 | grep "^{" | sed -n 2,2p > $O/i8_bench.txt
grep -A4 "B = 256" $O/marks.txt | tail -5; grep -A4 "B = 5000" $O/marks.txt | tail -5; cat $O/pytest_sel.txt $O/i8_bench.txt
