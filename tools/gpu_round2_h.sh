#!/bin/bash
mkdir -p gpurun_out/h
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/h/pytest_gpu.txt
timeout 300 python tools/api_phase_times.py > gpurun_out/h/phases.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/h/bench_n1.txt
tail -n 4 gpurun_out/h/pytest_gpu.txt; cat gpurun_out/h/phases.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/h/bench_n1.txt').read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('api_inclusive'))
PY
