import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import synthetic
from plspm import _native
C = synthetic.satisfaction_C()
X, blocks = synthetic.synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
def mk():
    m = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0); m.upload(X); return m
a, b = mk(), mk()
for B in (5000, 16384, 20000, 40000):
    a.bootstrap_device(B, seed=1)
    orig = np.zeros(a.row_width)
    t1, u1 = a.summary(B, orig)
    t2, u2 = a.summary(B, orig)
    rows = a.fetch(0, B)[0]
    print(B, "same-handle repeat equal:", np.array_equal(t1, t2), "max abs diff", np.abs(t1 - t2).max())
    b.bootstrap_device(B, seed=1)
    t3, _ = b.summary(B, orig)
    print(B, "other handle equal:", np.array_equal(t1, t3), np.abs(t1 - t3).max(), "cols differing", np.unique(np.nonzero(t1 != t3)[1]))
    print("   vs numpy mean", np.abs(t1[:, 1] - rows.mean(axis=0)).max(), "q", np.abs(t1[:, 3] - np.quantile(rows, 0.025, axis=0)).max())
