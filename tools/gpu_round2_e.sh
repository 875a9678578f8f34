#!/bin/bash
mkdir -p gpurun_out/e
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_seams.py -x -q 2>&1 | tail -6 > gpurun_out/e/pytest_parity.txt
timeout 600 python tools/c5_ab.py > gpurun_out/e/c5_ab.txt 2>&1
timeout 300 python tools/fit_bench.py c2 2>&1 | tail -1 > gpurun_out/e/fit_c2.txt
tail -n 3 gpurun_out/e/pytest_parity.txt; cat gpurun_out/e/c5_ab.txt gpurun_out/e/fit_c2.txt
