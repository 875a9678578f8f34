"""Schedule variants of gram_i8_kernel<7> (library built with EXTRA=-DPLSPM_I8_EXPERIMENTS): Gram time per variant and wave count."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from plspm import _native
from synthetic import satisfaction_C, synth
B = 5000
C = satisfaction_C()
X, blocks = synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
nm = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
nm.upload(X)
ref = None
for waves in (4, 8):
    for var in (0, 3, 6, 4, 12, 18, 21, 24, 30, 33):
        nm.set_option("i8_slices", 7); nm.set_option("i8_waves", waves); nm.set_option("i8_variant", var)
        rows = nm.bootstrap(64, seed=1)[0]
        if ref is None: ref = rows
        ok = bool(np.array_equal(rows, ref))
        for w in range(2): nm.bootstrap_device(B, seed=1, rep_offset=w * B)
        nm.sync(); nm.profile(True); nm.profile_reset()
        for k in range(10): nm.bootstrap_device(B, seed=1, rep_offset=(2 + k) * B)
        nm.sync(); nm.profile(False)
        ms, n = nm.profile_read("gram")
        print(json.dumps({"waves": waves, "variant": var, "pair_barrier": var >= 18, "NS": 5 if var >= 18 else 3 + var % 3, "rstep": 1 + (var // 3) % 3, "dma_head": (var % 18) // 9, "gram_ms": round(ms / n, 4), "identical_rows": ok}), flush=True)
