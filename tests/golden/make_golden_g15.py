#!/opt/conda/bin/python3.9
"""Golden g15 (build container only): higher order construct + missing values, fit and bootstrap rows from the REAL reference.

Run:   PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_golden_g15.py

The reference's two-stage estimate (estimator.py:29-55) on the mobi data of its own HOC test (tests/test_regression_seminr.py:49-74) with
NaNs punched into plain and constituent MVs (Scale.NUM: weights.py:88-98, mode.py:35-41), for the full sample and for explicit resample
index lists -- the rows a bootstrap worker forms (bootstrap.py:56-64).  Data only: inputs, indices, outputs."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import refshim  # noqa: E402

refshim.load_reference()
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import plspm.config as c  # noqa: E402
import plspm.weights as refw  # noqa: E402
import plspm.inner_model as refim  # noqa: E402
from plspm.estimator import Estimator  # noqa: E402
from plspm.mode import Mode  # noqa: E402
from plspm.scale import Scale  # noqa: E402
from plspm.scheme import Scheme  # noqa: E402

mobi = pd.read_csv(os.path.join(HERE, "ref_data", "mobi.csv"), index_col=0)
rs = np.random.RandomState(1515)
holes = mobi.copy().astype(float)          # (NaNs: the reference's Config.filter raises KeyError for a HOC model, config.py:279 -- nothing to reproduce)
g = {"data": holes.values, "columns": np.array(list(holes.columns)), "idx": rs.randint(250, size=(5, 250))}
for tag, scheme in (("path", Scheme.PATH), ("centroid", Scheme.CENTROID)):
    def hoc_config():
        st = c.Structure()
        st.add_path(["Expectation", "Quality"], ["Satisfaction"])
        st.add_path(["Satisfaction"], ["Complaints", "Loyalty"])
        cfg = c.Config(st.path(), default_scale=Scale.ORD)
        cfg.add_higher_order("Satisfaction", Mode.A, ["Image", "Value"])
        cfg.add_lv_with_columns_named("Expectation", Mode.A, holes, "CUEX")
        cfg.add_lv_with_columns_named("Quality", Mode.A, holes, "PERQ")
        cfg.add_lv_with_columns_named("Loyalty", Mode.A, holes, "CUSL")
        cfg.add_lv_with_columns_named("Image", Mode.A, holes, "IMAG")
        cfg.add_lv_with_columns_named("Complaints", Mode.A, holes, "CUSCO")
        cfg.add_lv_with_columns_named("Value", Mode.A, holes, "PERV")
        return cfg
    cfg = hoc_config()
    filtered = cfg.filter(holes)
    calc = refw.WeightsCalculatorFactory(cfg, 100, 1e-7, np.sqrt(250 / 249), scheme)
    rows, ok = [], []
    for idx in [np.arange(250)] + list(g["idx"]):
        est = Estimator(cfg)
        try:
            fd, sc, w = est.estimate(calc, filtered.iloc[idx, :])
            cfg2 = est.config()
            lvs2 = list(cfg2.path())
            mvs2 = [mv for lv in lvs2 for mv in cfg2.mvs(lv)]
            im = refim.InnerModel(cfg2.path(), sc)
            eff = im.effects()
            ld = (sc.apply(lambda s: fd.corrwith(s)) * cfg2.odm(cfg2.path())).sum(axis=1).loc[mvs2].values.astype(float)
            rows.append(np.concatenate((w.loc[mvs2, "weight"].values.astype(float), im.r_squared().loc[lvs2].values.astype(float),
                                        eff.loc[:, "total"].values.astype(float), eff.loc[:, "direct"].values.astype(float), ld)))
            ok.append(1)
            g[tag + "/lvs2"] = np.array(lvs2); g[tag + "/mvs2"] = np.array(mvs2)
            g[tag + "/eff_from"] = np.array([lvs2.index(x) for x in eff["from"]]); g[tag + "/eff_to"] = np.array([lvs2.index(x) for x in eff["to"]])
        except Exception as e:                                    # the reference's bootstrap drops such a replicate (bootstrap.py:65-66)
            print("replicate failed in the reference:", repr(e)[:200])
            rows.append(None); ok.append(0)
    width = max(len(r) for r in rows if r is not None)
    g[tag + "/rows"] = np.array([r if r is not None else np.full(width, np.nan) for r in rows])
    g[tag + "/ok"] = np.array(ok)
    print(tag, "rows", g[tag + "/rows"].shape, "ok", ok, "finite", np.isfinite(g[tag + "/rows"]).all(axis=1))
np.savez_compressed(os.path.join(HERE, "g15_hoc_ordinal.npz"), **g)
print("wrote g15_hoc_ordinal.npz")
