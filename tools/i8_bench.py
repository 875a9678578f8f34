"""Headline workload (10k x 60 x 6, 5,000 replicates per step) on the Gram variants: kernel times from the library's own HIP events.
Every configuration is measured in several interleaved rounds (the first measurement after a different kernel mix runs at another
clock: order effects of +-5 %); the table reports the minimum and the median per configuration."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from plspm import _native
from synthetic import satisfaction_C, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
C = satisfaction_C()
X, blocks = synth(10000, C, 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
configs = [(2, 0, 8, 16, 0), (1, 7, 4, 16, 0), (2, 7, 4, 16, 0), (2, 7, 4, 16, 1), (2, 7, 8, 16, 0), (2, 7, 8, 16, 1), (2, 7, 4, 32, 0), (2, 6, 4, 16, 1), (2, 8, 4, 16, 1), (2, 5, 4, 16, 1)]
models = {}
for cfg in configs:
    path, slices, waves, shape, sched = cfg
    nm = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0)
    nm.upload(X)
    nm.set_option("gram_path", path); nm.set_option("i8_slices", slices); nm.set_option("i8_waves", waves); nm.set_option("i8_shape", shape); nm.set_option("i8_sched", sched)
    for w in range(2):
        nm.bootstrap_device(B, seed=1, rep_offset=w * B)
    nm.sync()
    models[cfg] = nm
res = {cfg: [] for cfg in configs}
for rnd in range(ROUNDS):
    order = configs if rnd % 2 == 0 else configs[::-1]
    for cfg in order:
        nm = models[cfg]
        nm.profile(True); nm.profile_reset()
        t = time.perf_counter()
        steps = 10
        for k in range(steps):
            nm.bootstrap_device(B, seed=1, rep_offset=(2 + rnd * steps + k) * B)
        nm.sync()
        dt = (time.perf_counter() - t) / steps
        prof = {k: nm.profile_read(k) for k in _native.KERNELS}
        nm.profile(False)
        res[cfg].append((dt * 1e3, {k: v[0] / max(1, v[1]) for k, v in prof.items() if v[1]}))
for cfg in configs:
    steps_ms = [r[0] for r in res[cfg]]
    gram = [r[1]["gram"] for r in res[cfg]]
    print(json.dumps({"gram_path": cfg[0], "slices": cfg[1], "waves": cfg[2], "shape": cfg[3], "stream_k": cfg[4], "B": B, "ms_per_step_min": round(min(steps_ms), 4), "ms_per_step_median": round(float(np.median(steps_ms)), 4),
                      "replicates_per_s_best": round(B / min(steps_ms) * 1e3), "gram_ms_min": round(min(gram), 4), "gram_ms_median": round(float(np.median(gram)), 4),
                      "kernel_ms_last": {k: round(v, 4) for k, v in res[cfg][-1][1].items()}}))
