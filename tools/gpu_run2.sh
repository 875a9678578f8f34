#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu.txt
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -2 | tee gpurun_out/bench_n1.txt
