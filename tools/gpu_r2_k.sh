#!/bin/bash
mkdir -p gpurun_out/k
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/k/pytest_gpu.txt
cat gpurun_out/k/pytest_gpu.txt
