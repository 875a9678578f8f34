"""Where the wall time of a drop-in Plspm() call goes on the HOST (10k x 60 x 6, Scheme.PATH): cProfile of 30 calls without and 30 with bootstrap.
usage: api_pyprofile.py [calls]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import pandas as pd
import plspm.config as c
from plspm.mode import Mode
from plspm.plspm import Plspm
from plspm.scheme import Scheme
import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
X, blocks = synthetic.synth(10000, synthetic.satisfaction_C(), 10, seed=0)
lvs = synthetic.SAT_LVS
cols = ["%s%d" % (lv.lower(), k) for lv in lvs for k in range(10)]
frame = pd.DataFrame(X, columns=cols)
structure = c.Structure()
for frm, to in synthetic.SAT_EDGES: structure.add_path([frm], [to])
def config():
    cfg = c.Config(structure.path(), scaled=True)
    for lv in lvs: cfg.add_lv_with_columns_named(lv, Mode.A, frame, lv.lower())
    return cfg
for _ in range(3):
    Plspm(frame, config(), Scheme.PATH); Plspm(frame, config(), Scheme.PATH, bootstrap=True, bootstrap_iterations=5000, processes=1, seed=1)
for boot in (False, True):
    t0 = time.perf_counter()
    for _ in range(n):
        m = Plspm(frame, config(), Scheme.PATH, bootstrap=boot, bootstrap_iterations=5000, processes=1, seed=1)
    wall = (time.perf_counter() - t0) / n * 1e3
    pr = cProfile.Profile(); pr.enable()
    for _ in range(n):
        m = Plspm(frame, config(), Scheme.PATH, bootstrap=boot, bootstrap_iterations=5000, processes=1, seed=1)
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    print("==== bootstrap=%s: %.3f ms per call un-profiled" % (boot, wall)); print(s.getvalue()[:5500])
