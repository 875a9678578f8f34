"""Solver kernel time against the batch size (how the one-wave-per-problem kernels fill the 2,048 wave slots of the chip): kernel times
from the library's HIP events.  usage: solver_rounds.py [key=value ...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import synthetic as orc
from plspm import _native
X, blocks = orc.synth(10000, orc.satisfaction_C(), 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
for wave in (0, 1):
    m = _native.NativeModel(boff, orc.satisfaction_C().astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
    m.upload(X); m.set_option("solver_wave", wave)
    for o in sys.argv[1:]:
        k, v = o.split("="); m.set_option(k, int(v))
    out = {}
    for B in (256, 1024, 2048, 2304, 3072, 4096, 4352, 5000, 6144, 8192):
        for w in range(3): m.bootstrap_device(B, seed=1, rep_offset=w * B)
        m.sync(); m.profile(True, only="solver"); m.profile_reset()
        for w in range(10): m.bootstrap_device(B, seed=1, rep_offset=(3 + w) * B)
        m.sync()
        ms, n = m.profile_read("solver")
        out[B] = round(ms / max(1, n) * 1e3, 1)
        m.profile(False)
    print(json.dumps({"solver_wave": wave, "solver_us_by_B": out}))
