"""Workload of the counter passes of the int8 Gram (tools/gpu_profiles.sh): full-size launches (10k x 60 x 6, 5,000 replicates) of the default
gram_i8p_kernel and of the round-3 gram_i8_kernel (set_option i8_priv 0) in one process, so that ONE `rocprofv3 --pmc` pass yields both rows."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from plspm import _native
from synthetic import satisfaction_C, synth
C = satisfaction_C()
X, blocks = synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
nm = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
nm.upload(X)
for priv in (1, 0, 1, 0):
    nm.set_option("i8_priv", priv)
    for k in range(3):
        nm.bootstrap_device(5000, seed=1, rep_offset=k * 5000)
    nm.sync()
