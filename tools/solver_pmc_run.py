"""Five headline bootstrap steps with the wave solver and five with the rows solver (for rocprofv3 --pmc passes over both kernels)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import synthetic as orc
from plspm import _native
X, blocks = orc.synth(10000, orc.satisfaction_C(), 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
m = _native.NativeModel(boff, orc.satisfaction_C().astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
m.upload(X)
for o in sys.argv[1:]:
    k, v = o.split("="); m.set_option(k, int(v))
for wave in (1, 0):
    m.set_option("solver_wave", wave)
    for w in range(5): m.bootstrap_device(5000, seed=1, rep_offset=w * 5000)
    m.sync()
