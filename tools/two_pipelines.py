"""Experiment: the bootstrap steps dealt alternately to K handles (K streams, own buffers), enqueued without host synchronisation --
do the kernels of neighbouring steps fill each other's partial rounds (the Gram's fifth, the solver's third)?  usage: two_pipelines.py [B] [key=value ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import synthetic as orc
from plspm import _native
opts = [a for a in sys.argv[1:] if "=" in a]            # set_option pairs applied to every handle, e.g. i8_rt=8
args = [a for a in sys.argv[1:] if "=" not in a]
B = int(args[0]) if args else 5000
X, blocks = orc.synth(10000, orc.satisfaction_C(), 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
def mk():
    m = _native.NativeModel(boff, orc.satisfaction_C().astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
    m.upload(X)
    for o in opts:
        k, v = o.split("="); m.set_option(k, int(v))
    return m
models = [mk() for _ in range(3)]
for m in models:
    for w in range(3): m.bootstrap_device(B, seed=1, rep_offset=w * B)
    m.sync()
STEPS = 60
res = {1: [], 2: [], 3: []}
for rnd in range(5):
    for k in ((1, 2, 3) if rnd % 2 == 0 else (3, 2, 1)):
        t = time.perf_counter()
        for s in range(STEPS): models[s % k].bootstrap_device(B, seed=1, rep_offset=(3 + s) * B)
        for m in models[:k]: m.sync()
        res[k].append((time.perf_counter() - t) / STEPS * 1e3)
for k in (1, 2, 3):
    print(json.dumps({"handles": k, "B": B, "options": opts, "ms_per_step_min": round(min(res[k]), 4), "ms_per_step_median": round(float(np.median(res[k])), 4), "replicates_per_s_best": round(B / min(res[k]) * 1e3)}))
