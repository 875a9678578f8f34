#!/usr/bin/env python3
"""Non-metric (Scale.NUM) counterpart of the headline workload: 10k x 60 x 6, Mode A, PATH, B replicates per step.
Prints one JSON line with replicates/s and the per-kernel-class HIP-event times (solver = nm prepare/step/finish, scores = the
streaming convergence passes)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import synthetic as orc  # noqa: E402  (workload generator: data only)
from plspm import _native  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
steps = int(os.environ.get("NM_BENCH_STEPS", "20"))
NROWS = int(os.environ.get("NM_BENCH_N", "10000"))
X, blocks = orc.synth(NROWS, orc.satisfaction_C(), 10, seed=0)
boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
m = _native.NativeModel(boff, orc.satisfaction_C().astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0, nonmetric=True)
m.upload(X)
if "NM_BENCH_WAVE16" in os.environ: m.set_option("nm_wave16", int(os.environ["NM_BENCH_WAVE16"]))      # 0 = the per-iteration launches of rounds 1-5 (A/B)
for kv in filter(None, os.environ.get("NM_BENCH_OPTS", "").split(",")):      # any set_option key=value, e.g. NM_BENCH_OPTS=nm_verify_rows=1
    m.set_option(kv.split("=")[0], int(kv.split("=")[1]))
if "NM_BENCH_GRAM_PATH" in os.environ: m.set_option("gram_path", int(os.environ["NM_BENCH_GRAM_PATH"]))      # 1 = fp64 route (beyond 65,535 rows: row lists + gathering pass)
fit = m.fit(want_scores=False)
# as bench.py does for the headline: spin-up steps bring the device to its working clocks; the timed steps run un-profiled (a fresh
# replicate-id range every step), the per-kernel HIP-event times come from a second, profiled pass of the same steps
spin = int(os.environ.get("NM_BENCH_SPINUP", "60"))
for w in range(spin): m.bootstrap_device(B, seed=1, rep_offset=w * B)
m.sync()
t0 = time.perf_counter()
for w in range(steps): m.bootstrap_device(B, seed=1, rep_offset=(spin + w) * B)
m.sync()
dt = (time.perf_counter() - t0) / steps
m.profile(True); m.profile_reset()
t0 = time.perf_counter()
for w in range(steps):
    m.bootstrap_device(B, seed=1, rep_offset=(spin + steps + w) * B)
m.sync()
dt_prof = (time.perf_counter() - t0) / steps
m.profile(False)
k = {n: m.profile_read(n) for n in ("resample", "gram", "solver", "scores")}
rows, status, iters = m.bootstrap(64, seed=1)
print(json.dumps({"workload": "non-metric (Scale.NUM) %d x 60 x 6, Mode A, PATH, tol 1e-6, %d replicates per step" % (NROWS, B), "gram_path": m.get_option("last_gram_path"), "one_launch_solver": m.get_option("last_nm_wave16"), "flagged": m.get_option("last_nm_flagged"), "replayed": m.get_option("last_nm_replayed"),
                  "replicates_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3), "ms_per_step_profiled": round(dt_prof * 1e3, 3), "spinup_steps": spin, "fit_iterations": fit["iterations"],
                  "replicate_iterations": [int(iters.min()), int(iters.max())], "all_ok": bool(np.all(status == 0)),
                  "kernel_ms_per_step": {n: round(v[0] / steps, 3) for n, v in k.items()},
                  "launches_per_step": {n: v[1] // steps for n, v in k.items()}}))
