#!/bin/bash
for t in 64 128 256; do
PLSPM_SOLVER_THREADS=$t PLSPM_DEBUG_MARKS=1 timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep -E "solver clocks|last iter" | tail -2
PLSPM_SOLVER_THREADS=$t timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('threads $t', d['value'], d['ms_per_step'], d['kernels_ms_per_step'], d['roofline']['frac'])"
done
