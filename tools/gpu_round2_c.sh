#!/bin/bash
mkdir -p gpurun_out/c
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/c/pytest_gpu.txt
tail -n 5 gpurun_out/c/pytest_gpu.txt
