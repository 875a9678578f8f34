#!/bin/bash
mkdir -p gpurun_out/f
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/f/pytest_gpu.txt
timeout 600 python bench.py --no-cpu-baseline 2>gpurun_out/f/bench_n1.err | tail -1 > gpurun_out/f/bench_n1.txt
timeout 600 python tools/fit_bench.py 2>&1 | tail -2 > gpurun_out/f/fit_bench.txt
python - <<'PY' > gpurun_out/f/plspm_profile.txt 2>&1
import cProfile, pstats, sys, os, io, time
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
t0 = time.perf_counter()
for _ in range(20): m = Plspm(frame, cfg(), Scheme.PATH)
print("Plspm() wall ms", (time.perf_counter() - t0) / 20 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): m = Plspm(frame, cfg(), Scheme.PATH)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(25); print(s.getvalue()[:6000])
PY
tail -n 4 gpurun_out/f/pytest_gpu.txt; cut -c1-300 gpurun_out/f/bench_n1.txt; head -3 gpurun_out/f/plspm_profile.txt
