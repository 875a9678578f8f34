#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_run25; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_solver_quad.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python tools/solver_quad_ab.py 2>/dev/null | cut -c1-220 | grep 'solver_quad": 1'
timeout 600 python bench.py --no-cpu-baseline --no-api --no-next-rows 2>/dev/null | tail -1 > $O/bench_quick.json; python -c "
import json; d=json.load(open('$O/bench_quick.json')); print(d['value'], d['ms_per_step'], d.get('value_strict'), d['roofline'].get('avg_launch_ms'), d.get('cold'))"
