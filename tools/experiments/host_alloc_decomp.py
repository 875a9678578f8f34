import sys, time, os, json
sys.path.insert(0, "plspm-python_amd"); sys.path.insert(0, "tools")
import numpy as np
def t(f):
    t0 = time.perf_counter(); r = f(); return (time.perf_counter() - t0) * 1e3, r
def cycle():
    a_ms, a = t(lambda: np.empty((1000000, 20)))
    f_ms, _ = t(lambda: a.fill(0.0))
    f2_ms, _ = t(lambda: a.fill(1.0))
    def d():
        nonlocal a
        del a
    d_ms, _ = t(d)
    return [round(x, 2) for x in (a_ms, f_ms, f2_ms, d_ms)]
print("before hip:", [cycle() for _ in range(3)])
from plspm import _native
import synthetic as orc
C = np.zeros((2, 2), dtype=np.uint8); C[1, 0] = 1
X, blocks = orc.synth(1000, C, 3, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
m = _native.NativeModel(boff, C, np.zeros(2, dtype=np.int32), 1, True, 100, 1e-6, 0)
m.upload(X); m.fit()
print("after hip [alloc, first fill, second fill, free]:", [cycle() for _ in range(3)])
