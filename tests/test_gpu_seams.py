"""GPU: the two internal seams of the reference that this backend replaces, called exactly the way the reference
calls them and checked against golden g1 (made from the real reference):

  * WeightsCalculatorFactory(config, iterations, tolerance, correction, scheme).calculate(treated_data, path)
        -> (final_data, scores, weights)                reference plspm/weights.py:172-187, called from estimator.py:39,52
  * Estimator(config).estimate(calculator, filtered_data) -> (final_data, scores, weights) + Estimator.config()
        reference plspm/estimator.py:29-58, called from plspm.py:68 and bootstrap.py:57
Return shapes / labels per SURVEY.md 7 'ordering contracts': scores index = data.index, columns = list(path); weights index = MVs in
path-LV order, one column 'weight'."""
import numpy as np
import pandas as pd
import pytest

import plspm.config as c
from helpers import SAT_ADD_ORDER, SAT_PREFIX, assert_close, case_modes, load, satisfaction_frame
from plspm.estimator import Estimator
from plspm.mode import Mode
from plspm.scheme import Scheme
from plspm.weights import WeightsCalculatorFactory

import plspm_oracle as orc

pytestmark = pytest.mark.gpu
SCHEME = {"centroid": Scheme.CENTROID, "factorial": Scheme.FACTORIAL, "path": Scheme.PATH}
CASES = [("A", "centroid", 0), ("A", "path", 1), ("B", "factorial", 1), ("B", "path", 0), ("M", "centroid", 1)]


def sat_config(modes, scaled):
    sat = satisfaction_frame()
    s = c.Structure()
    s.add_path(["IMAG"], ["EXPE", "SAT", "LOY"]); s.add_path(["EXPE"], ["QUAL", "VAL", "SAT"])
    s.add_path(["QUAL"], ["VAL", "SAT"]); s.add_path(["VAL"], ["SAT"]); s.add_path(["SAT"], ["LOY"])
    cfg = c.Config(s.path(), scaled=bool(scaled))
    per_lv = dict(zip(orc.SAT_LVS, modes))
    for lv in SAT_ADD_ORDER:                                  # add_lv order != path order, as in the reference's own test
        cfg.add_lv_with_columns_named(lv, Mode.A if per_lv[lv] == "A" else Mode.B, sat, SAT_PREFIX[lv])
    return sat, cfg


def check_triple(final_data, scores, weights, data, cfg, gold, key):
    path = cfg.path()
    assert list(scores.columns) == list(path) and scores.index.equals(data.index)
    assert list(weights.columns) == ["weight"]
    assert list(weights.index) == [mv for lv in list(path) for mv in cfg.mvs(lv)]          # MVs in path-LV order (weights.py:31,69)
    names = list(gold[key + "/mv_names"])
    assert_close(weights.loc[names, "weight"].values, gold[key + "/weights"], 1e-8, what=key + " weights")
    assert_close(scores.loc[:, orc.SAT_LVS].values, gold[key + "/scores"], 1e-7, 1e-9, what=key + " scores")
    assert final_data.shape == (data.shape[0], len(names)) and list(final_data.columns) == names
    # final_data is the treated data: scores == final_data . W (weights.py:60) up to the sign rule
    W = pd.DataFrame(0.0, index=names, columns=list(path))
    for lv in list(path):
        for mv in cfg.mvs(lv):
            W.loc[mv, lv] = weights.loc[mv, "weight"]
    rebuilt = final_data.values @ W.values
    sign = np.sign((rebuilt * scores.values).sum(axis=0))
    assert_close(rebuilt * sign, scores.values, 1e-7, 1e-9, what=key + " scores from final_data and weights")


@pytest.mark.parametrize("modes,scheme,scaled", CASES)
def test_weights_calculator_factory_calculate_seam(modes, scheme, scaled):
    gold = load("g1_satisfaction")
    sat, cfg = sat_config(case_modes(modes), scaled)
    filtered = cfg.filter(sat)
    n = filtered.shape[0]
    treated = cfg.treat(filtered)                                     # estimator.py:33
    calculator = WeightsCalculatorFactory(cfg, 100, 1e-6, np.sqrt(n / (n - 1)), SCHEME[scheme])       # plspm.py:67
    final_data, scores, weights = calculator.clone().calculate(treated, cfg.path())                    # estimator.py:39
    check_triple(final_data, scores, weights, filtered, cfg, gold, "%s_%s_%d" % (modes, scheme, scaled))
    assert final_data is treated or np.array_equal(final_data.values, treated.values)
    assert calculator.config() is cfg


@pytest.mark.parametrize("modes,scheme,scaled", CASES)
def test_estimator_estimate_seam(modes, scheme, scaled):
    gold = load("g1_satisfaction")
    sat, cfg = sat_config(case_modes(modes), scaled)
    filtered = cfg.filter(sat)
    n = filtered.shape[0]
    calculator = WeightsCalculatorFactory(cfg, 100, 1e-6, np.sqrt(n / (n - 1)), SCHEME[scheme])
    estimator = Estimator(cfg)
    final_data, scores, weights = estimator.estimate(calculator, filtered)                             # plspm.py:68
    check_triple(final_data, scores, weights, filtered, cfg, gold, "%s_%s_%d" % (modes, scheme, scaled))
    assert_close(final_data.values, cfg.treat(filtered).values, 1e-12, 1e-14)
    assert estimator.config() is not cfg and list(estimator.config().path()) == list(cfg.path())      # works on a clone (estimator.py:30-31,53)
    # the bootstrap's use of the seam (bootstrap.py:57): a resampled frame with a repeated index
    idx = load("g4_satisfaction_boot")["idx"][0]
    fd, sc, w = estimator.estimate(calculator, filtered.iloc[idx, :])
    assert sc.shape == (n, 6) and sc.index.equals(filtered.index[idx]) and np.all(np.isfinite(w["weight"].values))


def test_calculate_raises_the_reference_exception_when_not_converging():
    sat, cfg = sat_config("AAAAAA", 0)
    filtered = cfg.filter(sat)
    n = filtered.shape[0]
    calculator = WeightsCalculatorFactory(cfg, 2, 1e-30, np.sqrt(n / (n - 1)), Scheme.CENTROID)
    with pytest.raises(Exception, match="Could not converge after 3 iterations"):                       # weights.py:185-186
        calculator.calculate(cfg.treat(filtered), cfg.path())
