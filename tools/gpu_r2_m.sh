#!/bin/bash
mkdir -p gpurun_out/m
timeout 900 python -m pytest tests/test_gpu_gram_i8.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/m/pytest_i8.txt
timeout 300 python tools/i8_bench.py > gpurun_out/m/i8_bench.txt 2>&1
cat gpurun_out/m/pytest_i8.txt gpurun_out/m/i8_bench.txt
