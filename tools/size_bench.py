"""Bootstrap throughput next to the headline size: N = 100,000 rows (the int8 route's resample counts from two
65,536-row windows per replicate) and 120 MVs / 12 LVs (7,381 pair columns; the quad solver -- round 5 -- behind the digit-plane Gram).  One JSON line
per workload: replicates/s, kernel times from the library's HIP events, the Gram route and solver taken."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import synthetic as orc
from plspm import _native


def chain_C(L):
    C = np.zeros((L, L), dtype=np.int64)
    for j in range(L):
        if j - 1 >= 0: C[j, j - 1] = 1
        if j - 3 >= 0: C[j, j - 3] = 1
    return C


for name, N, C, B in (("10k x 60 x 6 (headline)", 10000, orc.satisfaction_C(), 5000), ("100k x 60 x 6", 100000, orc.satisfaction_C(), 5000),
                      ("10k x 120 x 12", 10000, chain_C(12), 5000)):
    X, blocks = orc.synth(N, C, 10, seed=0)
    L = C.shape[0]
    boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
    m = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(L, dtype=np.int32), 2, True, 100, 1e-6, 0)
    m.upload(X)
    rows, st, it = m.bootstrap(64, seed=1)
    for w in range(40 if N <= 10000 else 6): m.bootstrap_device(B, seed=1, rep_offset=w * B)      # (brings the device to its working clocks)
    m.sync()
    t0 = time.perf_counter()
    for w in range(10): m.bootstrap_device(B, seed=1, rep_offset=(3 + w) * B)
    m.sync()
    wall = (time.perf_counter() - t0) / 10
    m.profile(True); m.profile_reset()
    for w in range(5): m.bootstrap_device(B, seed=1, rep_offset=(13 + w) * B)
    m.sync(); m.profile(False)
    k = {n: round(m.profile_read(n)[0] / max(1, m.profile_read(n)[1]), 4) for n in ("resample", "gram", "solver")}
    npair = (X.shape[1] + 1) * (X.shape[1] + 2) // 2
    S = m.get_option("last_i8_slices")
    ops = 2.0 * N * npair * S * B
    print(json.dumps({"workload": name, "replicates_per_step": B, "digit_planes": S, "replicates_per_s": round(B / wall, 1), "ms_per_step": round(wall * 1e3, 4), "kernels_ms": k,
                      "gram_path": m.get_option("last_gram_path"), "solver": m.get_option("last_solver"), "status_ok": bool(np.all(st == 0)),
                      "int8_TOPs_algorithmic": round(ops / (k["gram"] * 1e-3) / 1e12, 1)}), flush=True)
    m.close()
