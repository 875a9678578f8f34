#!/usr/bin/env python3
"""configs[4]'s scores download (1M x 20 doubles = 160 MB): wall of plspm_fit with the scores into a FRESH array (what Plspm() pays: first-touch page
faults of the destination) and into a re-used, already touched one (the copy alone).  Usage: python tools/experiments/scores_download.py"""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import synthetic as orc  # noqa: E402
from plspm import _native  # noqa: E402

n, L, per = 1000000, 20, 10
C = np.zeros((L, L), dtype=np.uint8)
for i in range(1, L):
    C[i, i - 1] = 1
X, blocks = orc.synth(n, C, per, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
m = _native.NativeModel(boff, C, np.ones(L, dtype=np.int32), 1, True, 100, 1e-6, 0)
m.upload(X)
m.fit(want_scores=True)
def med(f, k=5):
    t = []
    for _ in range(k):
        t0 = time.perf_counter(); f(); t.append((time.perf_counter() - t0) * 1e3)
    return round(float(np.median(t)), 3), [round(x, 2) for x in t]
buf = np.empty((n, L)); buf[:] = 0.0
ref = m.fit(want_scores=True)["scores"]
out = {"no_scores_ms": med(lambda: m.fit(want_scores=False)), "fresh_ms": med(lambda: m.fit(want_scores=True)), "reused_ms": med(lambda: m.fit(want_scores=True, scores_out=buf)),
       "reused_runtime_copy_ms": (m.set_option("upload_direct", 1), med(lambda: m.fit(want_scores=True, scores_out=buf)), m.set_option("upload_direct", 0))[1],
       "np_empty_plus_touch_ms": med(lambda: np.empty((n, L)).fill(0.0)), "identical": bool(np.array_equal(buf, ref))}
print(json.dumps(out))
