import json, os, sys, time
sys.path.insert(0, "plspm-python_amd"); sys.path.insert(0, "tools")
import numpy as np
import synthetic as orc
from plspm import _native
def chain_C(L):
    C = np.zeros((L, L), dtype=np.int64)
    for j in range(L):
        if j - 1 >= 0: C[j, j - 1] = 1
        if j - 3 >= 0: C[j, j - 3] = 1
    return C
C = chain_C(12); B = 5000
X, blocks = orc.synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
m = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(12, dtype=np.int32), 2, True, 100, 1e-6, 0)
m.upload(X)
ref = None
for thr in (128, 256, 64, 128, 256):
    m.set_option("solver_threads", thr)
    rows, st, it = m.bootstrap(64, seed=1)
    if ref is None: ref = rows
    for w in range(10): m.bootstrap_device(B, seed=1, rep_offset=w * B)
    m.sync(); m.profile(True); m.profile_reset()
    for w in range(5): m.bootstrap_device(B, seed=1, rep_offset=(13 + w) * B)
    m.sync(); m.profile(False)
    print(thr, round(m.profile_read("solver")[0] / max(1, m.profile_read("solver")[1]), 4), float(np.abs(rows - ref).max()), it.min(), it.max(), flush=True)
