#!/bin/bash
PLSPM_DEBUG_MARKS=1 timeout 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>&1 | grep -E "solver clocks|last iterate" | tail -2
