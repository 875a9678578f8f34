#!/bin/bash
mkdir -p gpurun_out/i
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/i/pytest_gpu.txt
timeout 600 python tools/nonmetric_bench.py 2>&1 | tail -1 > gpurun_out/i/nonmetric_bench.json
timeout 600 python tools/categorical_bench.py 2>&1 | tail -1 > gpurun_out/i/categorical_bench.json
tail -n 4 gpurun_out/i/pytest_gpu.txt; cat gpurun_out/i/nonmetric_bench.json gpurun_out/i/categorical_bench.json
