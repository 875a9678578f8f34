#!/usr/bin/env python3
"""Where the wall time of the drop-in API's bootstrap phase goes (10k x 60 x 6, 5,000 replicates): enqueue + kernels, device summary,
row download -- through the C-ABI wrappers, each phase host-synchronised."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import synthetic  # noqa: E402
from plspm import _native  # noqa: E402

C = synthetic.satisfaction_C()
X, blocks = synthetic.synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
B = 5000
res = {}
for label in ("cold handle", "warm handle"):
    m = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
    m.upload(X)
    m.fit(want_scores=False)
    orig = np.ones(m.row_width)
    reps = 1 if label == "cold handle" else 20
    t = {"bootstrap_device+sync": [], "summary": [], "fetch": []}
    for _ in range(reps):
        t0 = time.perf_counter(); m.bootstrap_device(B, seed=1); m.sync(); t1 = time.perf_counter()
        m.summary(B, orig); t2 = time.perf_counter()
        m.fetch(0, B); t3 = time.perf_counter()
        t["bootstrap_device+sync"].append(t1 - t0); t["summary"].append(t2 - t1); t["fetch"].append(t3 - t2)
    res[label] = {k: round(float(np.median(v)) * 1e3, 3) for k, v in t.items()}
    t0 = time.perf_counter(); m.close(); res[label]["close"] = round((time.perf_counter() - t0) * 1e3, 3)
t0 = time.perf_counter()
for _ in range(20):
    m = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
    m.close()
res["create+close_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
print(json.dumps(res))
# the API's pattern: a fresh handle per call (create, upload, fit, bootstrap, summary), previous one still alive
fresh = {"create+upload+fit": [], "bootstrap_device+sync": [], "summary": []}
prev = None
for _ in range(12):
    t0 = time.perf_counter()
    m = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
    m.upload(X); m.fit(want_scores=True, want_cov=True)
    t1 = time.perf_counter()
    m.bootstrap_device(B, seed=1); m.sync()
    t2 = time.perf_counter()
    m.summary(B, np.ones(m.row_width))
    t3 = time.perf_counter()
    fresh["create+upload+fit"].append(t1 - t0); fresh["bootstrap_device+sync"].append(t2 - t1); fresh["summary"].append(t3 - t2)
    prev = m
print(json.dumps({"fresh handle per call": {k: round(float(np.median(v[2:])) * 1e3, 3) for k, v in fresh.items()}}))
