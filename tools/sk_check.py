"""Persistent stream-K int8 Gram (set_option i8_sched 1) against the tiled launch: bit-identical rows at several batch sizes + step time."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from plspm import _native
from synthetic import satisfaction_C, synth
C = satisfaction_C()
X, blocks = synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
def make(sched):
    nm = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
    nm.upload(X); nm.set_option("i8_sched", sched); return nm
a, b = make(0), make(1)
for B in [int(v) for v in (sys.argv[1:] or ["16384", "64", "300", "5000", "8192", "40000"])]:
    ra, sa, ia = a.bootstrap(B, seed=3)
    t = time.perf_counter(); rb, sb, ib = b.bootstrap(B, seed=3); dt = time.perf_counter() - t
    bad = np.flatnonzero(~np.all((ra == rb) | (np.isnan(ra) & np.isnan(rb)), axis=1))
    for m in (a, b):
        for k in range(3): m.bootstrap_device(B, seed=1, rep_offset=k * B)
        m.sync()
    ts = []
    for m in (a, b):
        t = time.perf_counter()
        for k in range(10): m.bootstrap_device(B, seed=1, rep_offset=(3 + k) * B)
        m.sync(); ts.append((time.perf_counter() - t) / 10 * 1e3)
    print(json.dumps({"B": B, "identical": bool(bad.size == 0), "bad_rows": int(bad.size), "first_bad": [int(v) for v in bad[:8]], "status_sk": [int(v) for v in np.unique(sb)],
                      "ms_tiled": round(ts[0], 4), "ms_stream_k": round(ts[1], 4)}), flush=True)
