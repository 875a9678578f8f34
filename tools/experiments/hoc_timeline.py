#!/usr/bin/env python3
"""One 5,000-replicate bootstrap call of the mobi HOC model on ordinal items (the workload of tools/hoc_bench.py) for a kernel trace:
  rocprofv3 --kernel-trace -d /tmp/hp -o hp -- python tools/experiments/hoc_timeline.py;  python tools/experiments/cat_timeline.py /tmp/hp 400
(the last call's kernels are the tail of the trace)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "plspm-python_amd"))
import numpy as np
import pandas as pd
import plspm.config as c
import plspm.weights as w
from plspm.estimator import Estimator
from plspm.mode import Mode
from plspm.scale import Scale
from plspm.scheme import Scheme
mobi = pd.read_csv(os.path.join(ROOT, "tests", "golden", "ref_data", "mobi.csv"), index_col=0).astype(float)
structure = c.Structure()
structure.add_path(["Expectation", "Quality"], ["Satisfaction"])
structure.add_path(["Satisfaction"], ["Complaints", "Loyalty"])
config = c.Config(structure.path(), default_scale=Scale.ORD)
config.add_higher_order("Satisfaction", Mode.A, ["Image", "Value"])
for lv, prefix in (("Expectation", "CUEX"), ("Quality", "PERQ"), ("Loyalty", "CUSL"), ("Image", "IMAG"), ("Complaints", "CUSCO"), ("Value", "PERV")):
    config.add_lv_with_columns_named(lv, Mode.A, mobi, prefix)
observations = config.filter(mobi)
calculator = w.WeightsCalculatorFactory(config, 100, 1e-7, np.sqrt(250 / 249), Scheme.PATH, 0)
pair = Estimator(config).two_stage_bootstrap_handles(calculator, observations)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
for k in range(2):
    pair.native.bootstrap_device(B, seed=1, rep_offset=k * B)
pair.native.sync()
time.sleep(0.05)
t0 = time.perf_counter()
pair.native.bootstrap_device(B, seed=1, rep_offset=2 * B)
pair.native.sync()
print("call wall ms", round((time.perf_counter() - t0) * 1e3, 2))
