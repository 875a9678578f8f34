"""Phase times of the fit-only handle when fit-only and bootstrap calls alternate (the pattern of bench.py's api_inclusive loop): create / upload / fit / close
of the C-ABI handle, and the whole Plspm() calls."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from plspm import _native
import synthetic
C = synthetic.satisfaction_C()
X, blocks = synthetic.synth(10000, C, 10, seed=0)
boff = np.arange(0, 61, 10).astype(np.int32)
def mk(): return _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
T = {k: [] for k in ("create", "upload", "fit", "close", "b_create", "b_upload", "b_prepare", "b_fit", "b_boot", "b_summary", "b_close_prev")}
prev = None
for it in range(12):
    t = [time.perf_counter()]
    h = mk(); t.append(time.perf_counter())
    h.upload(X); t.append(time.perf_counter())
    h.fit(want_scores=True, want_cov=True); t.append(time.perf_counter())
    h.close(); t.append(time.perf_counter())
    hb = mk(); t.append(time.perf_counter())
    hb.upload(X); t.append(time.perf_counter())
    hb.prepare_bootstrap(); t.append(time.perf_counter())
    hb.fit(want_scores=True, want_cov=True); t.append(time.perf_counter())
    hb.bootstrap_device(5000, seed=1); t.append(time.perf_counter())
    hb.summary(5000, np.ones(hb.row_width)); t.append(time.perf_counter())
    if prev is not None: prev.close()
    t.append(time.perf_counter())
    prev = hb
    if it >= 2:
        for k, name in enumerate(T): T[name].append(t[k + 1] - t[k])
print(json.dumps({k: round(float(np.median(v)) * 1e3, 3) for k, v in T.items()}))
