#!/usr/bin/env python3
"""Experiment: headline Gram at one workgroup per CU (LDS-limited) vs two -- is there room to co-schedule the solver?"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import synthetic
from plspm import _native
C = synthetic.satisfaction_C()
X, blocks = synthetic.synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
m = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
m.upload(X)
for opts in ({}, {"gram_lds_kb": 90}, {"gram_lds_kb": 60}, {"solver_threads": 64}, {"solver_threads": 256}):
    for k, v in opts.items(): m.set_option(k, v)
    for _ in range(3): m.bootstrap_device(5000, seed=1)
    m.sync(); m.profile(True); m.profile_reset()
    t0 = time.perf_counter()
    for k in range(20): m.bootstrap_device(5000, seed=1, rep_offset=5000 * k)
    m.sync(); wall = (time.perf_counter() - t0) / 20
    ms = {n: round(m.profile_read(n)[0] / max(m.profile_read(n)[1], 1), 4) for n in ("resample", "gram", "solver")}
    m.profile(False)
    print(json.dumps({"opts": opts, "ms_per_step": round(wall * 1e3, 4), "kernels": ms}), flush=True)
    for k in opts: m.set_option(k, {"gram_lds_kb": 0, "solver_threads": 128}[k])
