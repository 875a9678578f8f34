"""Timing probes of the int8 Gram (library built with EXTRA=-DPLSPM_I8_EXPERIMENTS, path in PLSPM_HIP_LIB): the default schedule with its
LDS-DMA issue, its workgroup barrier and / or its fragment reads taken out (results are garbage; only the kernel time is read) -- what
each of them costs the matrix pipe."""
import json, os, sys
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
nm.set_option("i8_slices", 7)                 # the probes are instantiated for the seven-plane kernel
ref = nm.bootstrap(64, seed=1)[0]
names = {3: "default", 803: "LDS-DMA as buffer_load ... offen lds", 103: "no DMA issue", 203: "no barrier", 303: "no DMA, no barrier", 403: "no fragment reads", 503: "no DMA, no reads", 703: "MFMA stream only"}
for waves in (4, 8):
    for rnd in range(2):
        for var in (3, 803, 103, 203, 303, 403, 503, 703):
            nm.set_option("i8_waves", waves); nm.set_option("i8_variant", var)
            same = bool(np.array_equal(nm.bootstrap(64, seed=1)[0], ref)) if var in (3, 803) else None
            for w in range(3): nm.bootstrap_device(B, seed=1, rep_offset=w * B)
            nm.sync(); nm.profile(True, only="gram"); nm.profile_reset()
            for k in range(10): nm.bootstrap_device(B, seed=1, rep_offset=(3 + k) * B)
            nm.sync(); nm.profile(False)
            ms, n = nm.profile_read("gram")
            print(json.dumps({"waves": waves, "variant": var, "what": names[var], "gram_ms": round(ms / n, 4), "rows_identical": same}), flush=True)
