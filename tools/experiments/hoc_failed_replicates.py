"""Which replicates of the ordinal mobi HOC call (tools/hoc_bench.py's model) fail, with what status and after how many second-stage trips -- the stragglers behind the call's tail."""
import collections, os, sys
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
for k in range(4):
    pair.native.bootstrap_device(B, seed=1, rep_offset=k * B)
    pair.native.sync()
    rows, status, iters = pair.native.fetch(0, B)
    bad = np.nonzero(status != 0)[0]
    print("call", k, "failed", len(bad), "status/iters", [(int(status[b]), int(iters[b])) for b in bad], "iters of ok: max", int(iters[status == 0].max()),
          "hist of ok > 12:", dict(collections.Counter(int(i) for i in iters[status == 0] if i > 12)))
