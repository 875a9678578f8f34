#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/fit_bench.py c2 c5 2>&1 | tail -2 | tee gpurun_out/fit_bench.txt | python -c "
import sys,json
for l in sys.stdin: d=json.loads(l); print(d['config'], d['kernel_ms'], d['device_ms_total'], d['algorithmic'])"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
