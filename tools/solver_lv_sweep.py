"""Solver kernel time against the number of LVs at ~60 MVs (10k rows, Mode A, PATH, 5,000 replicates per step): which solver a model takes and what it costs.
One JSON line per model."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import synthetic as orc
from plspm import _native
from size_bench_models import chain_C

B = 5000
for L, per in ((6, 10), (8, 8), (10, 6), (12, 5), (16, 4), (12, 3), (20, 3)):
    C = chain_C(L)
    X, blocks = orc.synth(10000, C, per, seed=0)
    boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
    m = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(L, dtype=np.int32), 2, True, 100, 1e-6, 0)
    m.upload(X)
    rows, st, it = m.bootstrap(64, seed=1)
    for w in range(20): m.bootstrap_device(B, seed=1, rep_offset=w * B)
    m.sync()
    t0 = time.perf_counter()
    for w in range(10): m.bootstrap_device(B, seed=1, rep_offset=(3 + w) * B)
    m.sync()
    wall = (time.perf_counter() - t0) / 10
    m.profile(True); m.profile_reset()
    for w in range(5): m.bootstrap_device(B, seed=1, rep_offset=(13 + w) * B)
    m.sync(); m.profile(False)
    k = {n: round(m.profile_read(n)[0] / max(1, m.profile_read(n)[1]), 4) for n in ("resample", "gram", "solver")}
    print(json.dumps({"workload": "10k x %d x %d" % (L * per, L), "solver": m.get_option("last_solver"), "iterations": [int(it.min()), int(it.max())], "replicates_per_s": round(B / wall, 1),
                      "ms_per_step": round(wall * 1e3, 4), "kernels_ms": k}), flush=True)
    m.close()
