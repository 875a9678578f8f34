#!/bin/bash
mkdir -p gpurun_out/n
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -30 > gpurun_out/n/pytest_gpu.txt
timeout 600 python tools/nonmetric_bench.py 2>&1 | tail -1 > gpurun_out/n/nonmetric_bench.json
timeout 600 python tools/categorical_bench.py 2>&1 | tail -1 > gpurun_out/n/categorical_bench.json
timeout 300 python tools/i8_bench.py 2>&1 | sed -n 2,3p > gpurun_out/n/i8_bench.txt
cat gpurun_out/n/pytest_gpu.txt gpurun_out/n/nonmetric_bench.json gpurun_out/n/categorical_bench.json gpurun_out/n/i8_bench.txt
