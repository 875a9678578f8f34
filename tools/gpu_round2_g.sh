#!/bin/bash
mkdir -p gpurun_out/g
timeout 900 python -m pytest tests/test_gpu_seams.py tests/test_gpu_dist.py -x -q 2>&1 | tail -8 > gpurun_out/g/pytest.txt
timeout 300 python tools/api_phase_times.py > gpurun_out/g/phases.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/g/bench_n1.txt
tail -n 4 gpurun_out/g/pytest.txt; cat gpurun_out/g/phases.txt
python - <<'PY'
import json
d = json.loads(open('gpurun_out/g/bench_n1.txt').read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('api_inclusive'))
PY
