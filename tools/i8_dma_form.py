"""Gram-kernel time (HIP events around the kernel alone, 10 launches) of the two forms of the int8 Gram's LDS-DMA on the headline
workload: i8_dma = 1 (`global_load_lds_dwordx4`, 64-bit base per block) vs 2 (`buffer_load_dwordx4 ... offen lds`, per-workgroup
descriptors + 32-bit offsets, the default), alternating, at 4 and 8 waves."""
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
ref = nm.bootstrap(64, seed=1)[0]
for waves in (4, 8):
    for rnd in range(3):
        for form in (1, 2):
            nm.set_option("i8_waves", waves); nm.set_option("i8_dma", form)
            same = bool(np.array_equal(nm.bootstrap(64, seed=1)[0], ref))
            for w in range(3): nm.bootstrap_device(B, seed=1, rep_offset=w * B)
            nm.sync(); nm.profile(True, only="gram"); nm.profile_reset()
            for k in range(10): nm.bootstrap_device(B, seed=1, rep_offset=(3 + k) * B)
            nm.sync(); nm.profile(False)
            ms, n = nm.profile_read("gram")
            print(json.dumps({"waves": waves, "i8_dma": form, "used": nm.get_option("last_i8_dma"), "gram_ms": round(ms / n, 4), "rows_identical": same}), flush=True)
