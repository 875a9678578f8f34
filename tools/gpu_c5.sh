#!/bin/bash
mkdir -p gpurun_out
for nw in 4 8; do
echo "NW=$nw"; PLSPM_WIDE_NW=$nw timeout 900 python tools/fit_bench.py c5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], d['algorithmic']['TFLOPs_on_gram_time'])"
done
timeout 900 python -m pytest tests -m gpu -x -q -k "config5 or chain20 or tile_count or limits" 2>&1 | tail -2
