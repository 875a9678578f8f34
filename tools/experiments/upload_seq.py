"""Where an upload of configs[4] (1.6 GB) spends its time depending on what ran before it."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synthetic as orc
from plspm import _native
C = orc.chain_C(20) if hasattr(orc, "chain_C") else None
n = 1000000
X, blocks = orc.synth(n, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
m = _native.NativeModel(boff, C.astype(np.uint8), np.ones(20, dtype=np.int32), 1, True, 100, 1e-6, 0)
def t(label, f):
    t0 = time.time(); r = f(); print("%-28s %7.2f ms" % (label, (time.time() - t0) * 1e3), flush=True); return r
for direct in (0, 1, 0):
    m.set_option("upload_direct", direct)
    print("upload_direct", direct)
    for k in range(3): t("upload", lambda: m.upload(X))
    t("fit no scores", lambda: m.fit(want_scores=False))
    t("upload after fit(no scores)", lambda: m.upload(X))
    t("fit with scores", lambda: m.fit(want_scores=True))
    t("upload after fit(scores)", lambda: m.upload(X))
    t("upload", lambda: m.upload(X))
    t("fit with scores", lambda: m.fit(want_scores=True))
    t("fit with scores", lambda: m.fit(want_scores=True))
    t("upload after 2 fits", lambda: m.upload(X))
