"""Headline workload (10k x 60 x 6, 5,000 replicates per step) on both Gram paths: kernel times from the library's own HIP events."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from plspm import _native
from synthetic import satisfaction_C, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
C = satisfaction_C()
X, blocks = synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
names = ["resample", "gram", "solver", "scores", "pack", "reduce"]
for path, slices, waves in ((1, 7, 8), (2, 7, 8), (2, 7, 4), (2, 6, 8), (2, 8, 8), (2, 5, 8)):
    nm = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
    nm.upload(X)
    nm.set_option("gram_path", path); nm.set_option("i8_slices", slices); nm.set_option("i8_waves", waves)
    for w in range(3):
        nm.bootstrap_device(B, seed=1, rep_offset=w * B)
    nm.sync()
    nm.profile(True)
    t = time.perf_counter()
    steps = 20
    for k in range(steps):
        nm.bootstrap_device(B, seed=1, rep_offset=(3 + k) * B)
    nm.sync()
    dt = (time.perf_counter() - t) / steps
    prof = {k: nm.profile_read(k) for k in _native.KERNELS}
    nm.profile(False)
    print(json.dumps({"gram_path": path, "slices": slices, "waves": waves, "B": B, "ms_per_step_profiled": round(dt * 1e3, 4), "replicates_per_s": round(B / dt),
                      "kernel_ms": {k: round(v[0] / max(1, v[1]), 4) for k, v in prof.items() if v[1]}, "last_gram_path": nm.get_option("last_gram_path")}))
