#!/usr/bin/env python3
"""Experiment: do two handles (two HIP streams) overlap the latency-bound solver/resample kernels of one half-batch with
the MFMA-bound Gram kernel of the other?  Compares 1 x 5000 replicates on one stream with 2 x 2500 and 4 x 1250 on
concurrent streams."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import synthetic as orc
from plspm import _native

X, blocks = orc.synth(10000, orc.satisfaction_C(), 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
def mk():
    m = _native.NativeModel(boff, orc.satisfaction_C().astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
    m.upload(X); return m
models = [mk() for _ in range(4)]
def run(k, B=5000, steps=20):
    per = B // k
    for _ in range(3):
        for i in range(k): models[i].bootstrap_device(per, seed=1, rep_offset=i * per)
        for i in range(k): models[i].sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        for i in range(k): models[i].bootstrap_device(per, seed=1, rep_offset=i * per)
        for i in range(k): models[i].sync()
    dt = (time.perf_counter() - t0) / steps
    print("streams=%d  %.3f ms per %d replicates  -> %.0f rep/s" % (k, dt * 1e3, B, B / dt))
for k in (1, 2, 4):
    run(k)
