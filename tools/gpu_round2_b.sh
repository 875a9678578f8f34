#!/bin/bash
mkdir -p gpurun_out/b
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/b/pytest_gpu.txt
timeout 600 python bench.py --no-cpu-baseline 2>gpurun_out/b/bench_n1.err | tail -1 > gpurun_out/b/bench_n1.txt
timeout 300 python bench.py --group --no-cpu-baseline --no-api 2>gpurun_out/b/bench_group.err | tail -1 > gpurun_out/b/bench_n1_group.txt
python - <<'PY' > gpurun_out/b/plspm_profile.txt 2>&1
import cProfile, pstats, sys, os, io
sys.path.insert(0, 'plspm-python_amd'); sys.path.insert(0, 'tools')
import numpy as np, pandas as pd
import synthetic
import plspm.config as c
from plspm.mode import Mode
from plspm.plspm import Plspm
from plspm.scheme import Scheme
X, blocks = synthetic.synth(10000, synthetic.satisfaction_C(), 10, seed=0)
cols = ["%s%d" % (lv.lower(), k) for lv in synthetic.SAT_LVS for k in range(10)]
frame = pd.DataFrame(X, columns=cols)
st = c.Structure()
for a, b in synthetic.SAT_EDGES: st.add_path([a], [b])
def cfg():
    g = c.Config(st.path(), scaled=True)
    for lv in synthetic.SAT_LVS: g.add_lv_with_columns_named(lv, Mode.A, frame, lv.lower())
    return g
for _ in range(3): Plspm(frame, cfg(), Scheme.PATH)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): m = Plspm(frame, cfg(), Scheme.PATH)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:9000])
PY
tail -n 5 gpurun_out/b/pytest_gpu.txt; cat gpurun_out/b/bench_n1.txt | cut -c1-200
