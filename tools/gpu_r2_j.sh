#!/bin/bash
mkdir -p gpurun_out/j
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/j/pytest_gpu.txt
timeout 300 python tools/experiments/int_mm_ceiling.py > gpurun_out/j/int_mm.txt 2>&1
timeout 600 python tools/nonmetric_bench.py 2>&1 | tail -1 > gpurun_out/j/nonmetric_bench.json
timeout 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/j/bench.json
tail -n 4 gpurun_out/j/pytest_gpu.txt; cat gpurun_out/j/int_mm.txt gpurun_out/j/nonmetric_bench.json gpurun_out/j/bench.json
