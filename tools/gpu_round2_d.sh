#!/bin/bash
mkdir -p gpurun_out/d
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_seams.py -x -q 2>&1 | tail -15 > gpurun_out/d/pytest_parity.txt
timeout 600 python tools/fit_bench.py 2>&1 | tail -3 > gpurun_out/d/fit_bench.txt
tail -n 6 gpurun_out/d/pytest_parity.txt; cat gpurun_out/d/fit_bench.txt
