import cProfile, pstats, sys, os, io, time
sys.path.insert(0, 'plspm-python_amd'); sys.path.insert(0, 'tools')
import numpy as np, pandas as pd
import synthetic
import plspm.config as c
from plspm.mode import Mode
from plspm.plspm import Plspm
from plspm.scheme import Scheme
import plspm.bootstrap as pb
X, blocks = synthetic.synth(10000, synthetic.satisfaction_C(), 10, seed=0)
cols = ["%s%d" % (lv.lower(), k) for lv in synthetic.SAT_LVS for k in range(10)]
frame = pd.DataFrame(X, columns=cols)
st = c.Structure()
for a, b in synthetic.SAT_EDGES: st.add_path([a], [b])
def cfg():
    g = c.Config(st.path(), scaled=True)
    for lv in synthetic.SAT_LVS: g.add_lv_with_columns_named(lv, Mode.A, frame, lv.lower())
    return g
for _ in range(3): m = Plspm(frame, cfg(), Scheme.PATH, bootstrap=True, bootstrap_iterations=5000, processes=1, seed=1)
ts = []
for _ in range(10):
    m = Plspm(frame, cfg(), Scheme.PATH, bootstrap=True, bootstrap_iterations=5000, processes=1, seed=1)
    ts.append(m.timings())
print("fit ms", np.median([t["fit_s"] for t in ts]) * 1e3, "bootstrap ms", np.median([t["bootstrap_s"] for t in ts]) * 1e3)
pr = cProfile.Profile()
orig = pb.Bootstrap.__init__
def wrapped(self, *a, **k):
    pr.enable(); orig(self, *a, **k); pr.disable()
pb.Bootstrap.__init__ = wrapped
for _ in range(10): m = Plspm(frame, cfg(), Scheme.PATH, bootstrap=True, bootstrap_iterations=5000, processes=1, seed=1)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14); print(s.getvalue()[:3500])
