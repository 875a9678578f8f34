#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 900 python tools/fit_bench.py c2 c5 2>&1 | tail -2 | tee gpurun_out/fit_bench.txt | python -c "
import sys,json
for l in sys.stdin: d=json.loads(l); print(d['config'], d['kernel_ms'], d['device_ms_total'], 'wall(no scores)', d['fit_wall_ms_no_scores'], 'wall', d['fit_wall_ms_incl_scores_download'])"
