#!/opt/conda/bin/python3.9
"""Build container only: the REAL reference on the case of tests/test_gpu_nmx.py::test_a_replicate_that_can_never_converge_... -- a Scale.NUM column that is constant in a
resample.  Run:  PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/probe_num_constant_replicate.py
Output (plspm 0.5.6): replicates 0 and 2 estimate (5 / 6 iterations), replicates 1 and 3 raise MissingDataError("exog contains inf or nans") -- BootstrapProcess.run drops them."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT,"tests","golden")); sys.path.insert(0, os.path.join(ROOT,"tests"))
warnings.filterwarnings("ignore")
import make_golden as mg
import numpy as np, pandas as pd
import plspm.config as c
from plspm.mode import Mode
from plspm.scale import Scale
import plspm_oracle as orc
C = orc.chain_C(3)
X, blocks = orc.synth(120, C, 3, seed=44)
X[:, 0] = 0.0; X[:4, 0] = 1.0
rs = np.random.RandomState(3)
idx = np.vstack([np.arange(120), 4 + rs.randint(116, size=120), rs.randint(120, size=120), 4 + rs.randint(116, size=120)])
lvs = ["L0","L1","L2"]; names = ["x%d" % p for p in range(9)]
df = pd.DataFrame(X, columns=names)
cfg = c.Config(mg.path_frame(C, lvs), scaled=True, default_scale=Scale.NUM)
for l in range(3): cfg.add_lv(lvs[l], Mode.A, *[c.MV(names[p], Scale.NUM) for p in blocks[l]])
from plspm.plspm import Plspm
m = Plspm(df, cfg, mg.SCHEMES["path"], 100, 1e-6)
eff = list(m.effects().index)
for k in range(4):
    try:
        rows, its = mg.boot_rows(df, cfg, "path", lvs, [idx[k]], eff, tol=1e-6)
        print(k, "reference ok", its, np.isfinite(rows).all(), rows[0][:3])
    except Exception as e:
        print(k, "reference raises", type(e).__name__, str(e)[:80])
