#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/pytest_gpu.txt
PLSPM_DEBUG_MARKS=1 timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep -E "solver clocks|last iterate|plspm cov" | tail -3
timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_n1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernels_ms_per_step'], d['roofline']['frac'])"
