"""A/B of the two solvers of 65 ... 128-MV Mode-A models on the 10k x 120 x 12 workload of tools/size_bench.py (and an 8 x 16 one): the quad solver
(solver_quad.h, option solver_quad 1) against the split rows solver (solver_quad 0).  One JSON line per (workload, solver): replicates/s and the
kernel times from the library's HIP events."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import synthetic as orc
from plspm import _native
from size_bench_models import chain_C

B = 5000
for name, L, per, scheme in (("10k x 120 x 12 path", 12, 10, 2), ("10k x 128 x 16 path", 16, 8, 2), ("10k x 120 x 12 centroid", 12, 10, 0), ("10k x 100 x 4 factorial", 4, 25, 1)):
    C = chain_C(L)
    X, blocks = orc.synth(10000, C, per, seed=0)
    boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
    m = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(L, dtype=np.int32), scheme, True, 100, 1e-6, 0)
    m.upload(X)
    for rounds in range(2):
        for quad in (1, 0):
            m.set_option("solver_quad", quad)
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
            print(json.dumps({"workload": name, "solver_quad": quad, "solver": m.get_option("last_solver"), "replicates_per_s": round(B / wall, 1), "ms_per_step": round(wall * 1e3, 4),
                              "kernels_ms": k}), flush=True)
    m.close()
