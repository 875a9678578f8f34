#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (build container only): the conversion factor SURVEY.md 8(d) asks for -- the NumPy oracle (the `cpu_baseline` of
bench.py, kind "port") and the REAL reference timed back to back on THIS container's cores on BASELINE.json configs[2] (10k x 60 x 6,
Mode A, Scheme.PATH, scaled), so that the oracle's replicates/s measured on a GPU box converts to reference-equivalent replicates/s.
The reference runs under /opt/conda/bin/python3.9 (oracle/refshim.py), the oracle under the interpreter bench.py uses.
Writes profiles/<tag>_cpu_calibration.json.   Run: python oracle/time_calibration.py [reference replicates, default 400 -- SURVEY.md 8(d) asks for >= 400] [tag, default r05]"""
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

REF_CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %(here)r)
for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"): os.environ[k] = "1"
import refshim
refshim.load_reference()
import pandas as pd
import plspm.config as c
from plspm.mode import Mode
from plspm.plspm import Plspm
from plspm.scheme import Scheme
import plspm_oracle as orc
reps = int(sys.argv[1])
X, blocks = orc.synth(10000, orc.satisfaction_C(), 10, seed=0)
names = ["x%%d" %% i for i in range(60)]
df = pd.DataFrame(X, columns=names)
lvs = orc.SAT_LVS
path = pd.DataFrame(orc.satisfaction_C(), index=lvs, columns=lvs)
runs = []
for procs, n in ((8, reps), (1, max(10, reps // 4 if reps < 400 else 100))):
    cfg = c.Config(path, scaled=True)
    for lv, b in zip(lvs, blocks): cfg.add_lv(lv, Mode.A, *[c.MV(names[i]) for i in b])
    t0 = time.time(); Plspm(df, cfg, Scheme.PATH); t_fit = time.time() - t0
    t0 = time.time(); Plspm(df, cfg, Scheme.PATH, bootstrap=True, bootstrap_iterations=n, processes=procs); t_all = time.time() - t0
    runs.append({"processes": procs, "replicates": n, "wall_s": round(t_all, 2), "single_fit_s": round(t_fit, 3), "replicates_per_s": round(n / (t_all - t_fit), 3)})
print(json.dumps(runs))
'''


def oracle_rates():
    sys.path.insert(0, ROOT)
    import bench                                        # the very harness of bench.py's cpu_baseline leg
    single = 8 / bench.cpu_worker((1, 8))
    pool = bench.cpu_baseline(budget_s=20.0)
    return single, pool


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    tag = sys.argv[2] if len(sys.argv) > 2 else "r05"
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    t0 = time.time()
    out = subprocess.run(["/opt/conda/bin/python3.9", "-c", REF_CHILD % {"here": HERE}, str(reps)], env=env, capture_output=True, text=True, check=True)
    ref_runs = json.loads(out.stdout.strip().splitlines()[-1])
    single, pool = oracle_rates()
    ref8 = [r for r in ref_runs if r["processes"] == 8][0]["replicates_per_s"]
    ref1 = [r for r in ref_runs if r["processes"] == 1][0]["replicates_per_s"]
    res = {"config": "10k x 60 x 6, Mode A, Scheme.PATH, scaled (BASELINE.json configs[2])",
           "host": {"cpus": os.cpu_count(), "model": next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")},
           "reference": {"version": "plspm 0.5.6 via its public API (oracle/refshim.py, /opt/conda/bin/python3.9)", "runs": ref_runs},
           "oracle": {"interpreter": sys.version.split()[0], "single_process_replicates_per_s": round(single, 2), "pool": pool},
           "oracle_over_reference": {"all_cores": round(pool["value"] / ref8, 2), "single_process": round(single / ref1, 2),
                                     "note": "divide a GPU box's cpu_baseline.value (oracle, one worker per core) by `all_cores` for the reference's "
                                             "replicates/s on that box's cores (same arithmetic, pandas / statsmodels overheads included)"},
           "wall_s": round(time.time() - t0, 1)}
    with open(os.path.join(ROOT, "profiles", tag + "_cpu_calibration.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
