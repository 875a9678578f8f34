"""Interleaved A/B of set_option(key, value) pairs on the headline workload (10k x 60 x 6, 5,000 replicates per step): wall time of
20 un-profiled steps per measurement, several rounds in alternating order.  usage: aux_ab.py key=v0,v1 [key2=...] [B]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from plspm import _native
from synthetic import satisfaction_C, synth
opts = [a for a in sys.argv[1:] if "=" in a]
B = int([a for a in sys.argv[1:] if "=" not in a][0]) if [a for a in sys.argv[1:] if "=" not in a] else 5000
key, vals = opts[0].split("="); vals = [int(v) for v in vals.split(",")]
C = satisfaction_C()
X, blocks = synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
models = {}
for v in vals:
    nm = _native.NativeModel(boff, C.astype(np.uint8), np.array([1 if c == "B" else 0 for c in os.environ.get("AB_MODES", "AAAAAA")], dtype=np.int32), 2, True, 100, 1e-6, 0)
    nm.upload(X)
    nm.set_option(key, v)
    for o in opts[1:]:
        k2, v2 = o.split("="); nm.set_option(k2, int(v2))
    for w in range(3): nm.bootstrap_device(B, seed=1, rep_offset=w * B)
    nm.sync(); models[v] = nm
solver_ms = {}
res = {v: [] for v in vals}
for rnd in range(6):
    for v in (vals if rnd % 2 == 0 else vals[::-1]):
        nm = models[v]
        t = time.perf_counter()
        for k in range(20): nm.bootstrap_device(B, seed=1, rep_offset=(3 + rnd * 20 + k) * B)
        nm.sync()
        res[v].append((time.perf_counter() - t) / 20 * 1e3)
        nm.profile(True, only="solver"); nm.profile_reset()
        for k in range(5): nm.bootstrap_device(B, seed=1, rep_offset=(900 + k) * B)
        nm.sync(); nm.profile(False)
        ms, n = nm.profile_read("solver"); solver_ms.setdefault(v, []).append(ms / max(n, 1))
ref = None
for v in vals:
    rows, st, it = models[v].bootstrap(64, seed=9)
    if ref is None: ref = rows
    print(json.dumps({key: v, "B": B, "ms_per_step_min": round(min(res[v]), 4), "ms_per_step_median": round(float(np.median(res[v])), 4),
                      "replicates_per_s_best": round(B / min(res[v]) * 1e3), "solver_ms_min": round(min(solver_ms[v]), 4), "last_solver": models[v].get_option("last_solver"),
                      "modes": os.environ.get("AB_MODES", "AAAAAA"), "rows_equal_first": bool(np.array_equal(rows, ref)),
                      "max_rel_row_difference_vs_first": float(np.max(np.abs(rows - ref) / np.maximum(np.abs(ref), 1e-6)))}))
