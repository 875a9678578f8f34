#!/opt/conda/bin/python3.9
"""TEST INFRASTRUCTURE (build container only): time the REAL reference's bootstrap on BASELINE.json configs[2]
(10k x 60 x 6, Mode A, Scheme.PATH, scaled) through its public API, processes = 1 and 8, >= 400 replicates each.
Writes profiles/r01_reference_cpu_timing.json.  Run: PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 oracle/time_reference.py
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[k] = "1"
import refshim  # noqa: E402

refshim.load_reference()
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import plspm.config as c  # noqa: E402
from plspm.mode import Mode  # noqa: E402
from plspm.plspm import Plspm  # noqa: E402
from plspm.scheme import Scheme  # noqa: E402
import plspm_oracle as orc  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    X, blocks = orc.synth(10000, orc.satisfaction_C(), 10, seed=0)
    names = ["x%d" % i for i in range(60)]
    df = pd.DataFrame(X, columns=names)
    lvs = orc.SAT_LVS
    path = pd.DataFrame(orc.satisfaction_C(), index=lvs, columns=lvs)
    out = {"config": "10k x 60 x 6, Mode A, Scheme.PATH, scaled, reference v0.5.6 via public API", "host_cpus": os.cpu_count(), "runs": []}
    for procs in (8, 1):
        cfg = c.Config(path, scaled=True)
        for lv, b in zip(lvs, blocks):
            cfg.add_lv(lv, Mode.A, *[c.MV(names[i]) for i in b])
        t0 = time.time()
        Plspm(df, cfg, Scheme.PATH)
        t_fit = time.time() - t0
        t0 = time.time()
        Plspm(df, cfg, Scheme.PATH, bootstrap=True, bootstrap_iterations=reps, processes=procs)
        t_all = time.time() - t0
        run = {"processes": procs, "replicates": reps, "wall_s": round(t_all, 2), "single_fit_s": round(t_fit, 3),
               "replicates_per_s": round(reps / (t_all - t_fit), 3)}
        print(run, flush=True)
        out["runs"].append(run)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r01_reference_cpu_timing.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
