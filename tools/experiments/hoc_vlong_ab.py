"""A/B of option nm_vlong (one long verification round for the stragglers of the one-launch categorical batch) on the ordinal mobi HOC call: ms per 5,000 replicates, alternating."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("plspm-python_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, pandas as pd
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
B = 5000
for k in range(2): pair.native.bootstrap_device(B, seed=1, rep_offset=k * B)
pair.native.sync()
for rnd in range(3):
    for v in (0, 1):
        pair.native.set_option("nm_vlong", v)
        t0 = time.perf_counter()
        for k in range(4): pair.native.bootstrap_device(B, seed=1, rep_offset=(2 + k) * B)
        pair.native.sync()
        print("nm_vlong", v, "ms per 5000: %.2f" % ((time.perf_counter() - t0) / 4 * 1e3))
