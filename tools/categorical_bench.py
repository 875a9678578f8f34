#!/usr/bin/env python3
"""Categorical (Scale.ORD, 5-point Likert) counterpart of the headline workload: 10k x 60 x 6, Mode A, PATH, B replicates per
step -- 300 indicator columns on the device.  Prints one JSON line with replicates/s and per-kernel-class HIP-event times."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import synthetic  # noqa: E402
from plspm import _native  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps = int(os.environ.get("CAT_BENCH_STEPS", "3"))
C = synthetic.satisfaction_C()
X, blocks = synthetic.synth(10000, C, 10, seed=0)
Z = (X - X.mean(axis=0)) / X.std(axis=0)
likert = np.clip(np.round(3 + 1.1 * Z), 1, 5)
cols, mv_off, boff = [], [0], [0]
for b in blocks:
    for p in b:
        codes = likert[:, p].astype(int) - 1
        ind = np.zeros((10000, 5)); ind[np.arange(10000), codes] = 1.0
        cols.append(ind); mv_off.append(mv_off[-1] + 5)
    boff.append(mv_off[-1])
Xaug = np.ascontiguousarray(np.concatenate(cols, axis=1))
m = _native.NativeModel(np.array(boff, dtype=np.int32), C.astype(np.uint8), np.zeros(6, dtype=np.int32), 2, True, 100, 1e-6, 0, nonmetric=True,
                        categorical=(np.array(mv_off, dtype=np.int32), np.ones(60, dtype=np.int32)))
m.upload(Xaug)
for kv in filter(None, os.environ.get("CAT_BENCH_OPTS", "").split(",")):      # any set_option key=value (A/B runs), e.g. CAT_BENCH_OPTS=nm_subset=0
    m.set_option(kv.split("=")[0], int(kv.split("=")[1]))
if "CAT_NM_WAVE" in os.environ: m.set_option("nm_wave", int(os.environ["CAT_NM_WAVE"]))        # A/B: 0 = the workgroup step of rounds 2-4 (nmg_kernel<1>)
if "CAT_NM_MFMA" in os.environ: m.set_option("nm_mfma", int(os.environ["CAT_NM_MFMA"]))      # A/B: 0 = the stop-rule pass on category codes (LDS lookups) instead of the int8 matrix product
if "CAT_NM_DIRECT16" in os.environ: m.set_option("nm_direct16", int(os.environ["CAT_NM_DIRECT16"]))      # A/B: 0 = packed fp64 moment matrices + the scatter pass (nmg_kernel<3>)
if "CAT_CONV_GY" in os.environ: m.set_option("conv_gy", int(os.environ["CAT_CONV_GY"]))      # waves the matrix-product pass aims at, in units of 256
if "CAT_NM_CODES" in os.environ: m.set_option("nm_codes", int(os.environ["CAT_NM_CODES"]))      # A/B: 0 = the stop-rule pass as multiply-adds over the 0/1 columns
t0 = time.perf_counter(); fit = m.fit(want_scores=False); t_fit = time.perf_counter() - t0
m.bootstrap_device(B, seed=1); m.sync()
m.profile(True); m.profile_reset()
t0 = time.perf_counter()
for _ in range(steps):
    m.bootstrap_device(B, seed=1)
    m.sync()
dt = (time.perf_counter() - t0) / steps
rows, status, iters = m.bootstrap(32, seed=1)
k = {n: m.profile_read(n) for n in ("resample", "gram", "solver", "scores")}
print(json.dumps({"workload": "categorical (Scale.ORD, 5-point) 10k x 60 MVs (300 indicator columns) x 6, Mode A, PATH, %d replicates per step" % B,
                  "replicates_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3), "step_kernel": ("nmw_step_kernel (one wave per problem)" if m.get_option("last_nm_wave") else "nmg_kernel<1> (workgroup)"), "stop_rule_pass": ("int8 matrix product (indicator bytes x digit planes of the score maps)" if m.get_option("last_nm_mfma") else "category codes" if m.get_option("last_nm_codes") else "multiply-adds over the indicator columns"), "fit_iterations": fit["iterations"], "fit_status": fit["status"],
                  "fit_wall_ms": round(t_fit * 1e3, 2), "replicate_iterations": [int(iters.min()), int(iters.max())], "all_ok": bool(np.all(status == 0)),
                  "kernel_ms_per_step": {n: round(v[0] / steps, 3) for n, v in k.items()}}))
