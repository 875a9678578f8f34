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


# ---------------------------------------------------------------------------------------------- operator plug-ins
def _reference_scheme(name, path, y):
    """NumPy restatement of reference scheme.py:27-28 / 36-37 / 45-54 (test-side checker)."""
    C = path.values
    if name == "centroid":
        return np.sign(np.corrcoef(y, rowvar=False) * (C + C.T))
    if name == "factorial":
        return np.cov(y, rowvar=False) * (C + C.T)
    E = C.astype(np.float64)
    for i in range(C.shape[0]):
        follow = C[i, :] == 1
        if follow.any():
            E[follow, i] = np.linalg.pinv(y[:, follow]) @ y[:, i]
        predec = C[:, i] == 1
        if predec.any():
            E[predec, i] = np.corrcoef(np.column_stack((y[:, predec], y[:, i])), rowvar=False)[:, -1][:-1]
    return E


@pytest.mark.parametrize("name", ["centroid", "factorial", "path"])
@pytest.mark.parametrize("centred", [True, False])
def test_scheme_operator_calculate(name, centred):
    """Scheme.X.value.calculate(path, y) (reference scheme.py:27,36,45) as a device call, for standardised scores (how the solver
    calls it, weights.py:45) and for arbitrary ones (uncentred: the no-intercept OLS of the path scheme sees the raw moments)."""
    sat, cfg = sat_config("AAAAAA", 1)
    path = cfg.path()
    rs = np.random.RandomState(4)
    eta = rs.standard_normal((500, 6)) @ (np.eye(6) + 0.5 * np.tril(rs.standard_normal((6, 6)), -1)).T
    y = (eta - eta.mean(axis=0)) / eta.std(axis=0, ddof=1) if centred else eta * np.array([1, 2, 0.5, 3, 1, 0.1]) + np.array([0.3, -2, 5, 0, 1, -0.4])
    got = SCHEME[name].value.calculate(path, y)
    want = _reference_scheme(name, path, y)
    if name == "path":
        assert isinstance(got, np.ndarray)
    else:
        assert isinstance(got, pd.DataFrame) and list(got.index) == list(path.index) and list(got.columns) == list(path.columns)
    assert_close(np.asarray(got), want, 1e-9, 1e-12, what=name)
    assert_close(y @ np.asarray(got), y @ want, 1e-9, 1e-10)                            # Z = Y E (weights.py:46)


@pytest.mark.parametrize("mode", ["A", "B"])
def test_mode_operator_outer_weights_metric(mode):
    """Mode.X.value.outer_weights_metric(data, Z, lv, mvs) (reference mode.py:28,50): k x 1 DataFrame, index = mvs, column = lv."""
    sat, cfg = sat_config("AAAAAA", 1)
    data = cfg.treat(cfg.filter(sat))
    rs = np.random.RandomState(8)
    Z = pd.DataFrame(rs.standard_normal((data.shape[0], 6)), index=data.index, columns=orc.SAT_LVS)
    mvs = list(cfg.mvs("SAT"))
    got = (Mode.A if mode == "A" else Mode.B).value.outer_weights_metric(data, Z, "SAT", mvs)
    assert isinstance(got, pd.DataFrame) and list(got.index) == mvs and list(got.columns) == ["SAT"]
    X = data.loc[:, mvs].values
    z = Z.loc[:, "SAT"].values
    want = X.T @ z / X.shape[0] if mode == "A" else np.linalg.lstsq(X, z, rcond=None)[0]
    assert_close(got["SAT"].values, want, 1e-9, 1e-12, what="mode " + mode)


def _reference_outer_weights_nonmetric(mode, X, present, z, correction):
    """NumPy restatement of reference mode.py:31-42 / 54-61 + util.treat_numpy (util.py:43-53) (test-side checker)."""
    if mode == "A" and present is not None:
        w = np.nansum(X * z[:, None], axis=0) / np.sum(np.power(present * z[:, None], 2), axis=0)
        Y = np.nansum(X.T * w[:, None], axis=0) / np.sum(np.power(present.T * w[:, None], 2), axis=0)
    elif mode == "A":
        w = X.T @ z / np.power(z, 2).sum()
        Y = X @ w
    else:
        w = np.linalg.lstsq(X, z, rcond=None)[0]
        Y = X @ w
    Y = Y - np.nanmean(Y)
    return w, Y / np.nanstd(Y, axis=0, ddof=1) * correction


@pytest.mark.parametrize("mode,missing", [("A", False), ("A", True), ("B", False)])
def test_mode_operator_outer_weights_nonmetric(mode, missing):
    """Mode.X.value.outer_weights_nonmetric(mv_grouped_by_lv, mv_grouped_by_lv_missing, Z, lv, correction) (reference mode.py:31-42,
    54-61) as one device call: (weights, Y) with Y = treat_numpy(X w) * correction; NaN-aware ratios when the LV's block has missing
    cells (the reference keeps the LV's presence mask only then); Mode B refuses missing data with the reference's message."""
    rs = np.random.RandomState(12)
    N, k = 777, 5
    z = rs.standard_normal(N)
    X = 0.6 * z[:, None] * np.linspace(0.5, 1.0, k) + rs.standard_normal((N, k))
    X = (X - X.mean(axis=0)) / X.std(axis=0)
    groups, masks = {"SAT": X.copy()}, {}
    if missing:
        holes = rs.rand(N, k) < 0.04
        holes[holes.all(axis=1)] = False
        groups["SAT"][holes] = np.nan
        masks["SAT"] = 1 - np.isnan(groups["SAT"])
    corr = np.sqrt(N / (N - 1))
    w, Y = (Mode.A if mode == "A" else Mode.B).value.outer_weights_nonmetric(groups, masks, z, "SAT", corr)
    want_w, want_Y = _reference_outer_weights_nonmetric(mode, groups["SAT"], masks.get("SAT"), z, corr)
    assert isinstance(w, np.ndarray) and w.shape == (k,) and Y.shape == (N,)
    assert_close(w, want_w, 1e-10, 1e-13, what="weights " + mode)
    assert_close(Y, want_Y, 1e-9, 1e-12, what="scores " + mode)
    if missing:
        with pytest.raises(Exception, match="not supported in mode B"):
            Mode.B.value.outer_weights_nonmetric(groups, masks, z, "SAT", corr)


def test_mode_b_operator_minimum_norm_on_a_collinear_block():
    sat, cfg = sat_config("AAAAAA", 0)
    data = cfg.treat(cfg.filter(sat))
    data = data.assign(imag1dup=data["imag1"])
    mvs = list(cfg.mvs("IMAG")) + ["imag1dup"]
    Z = pd.DataFrame({"IMAG": np.random.RandomState(1).standard_normal(data.shape[0])}, index=data.index)
    got = Mode.B.value.outer_weights_metric(data, Z, "IMAG", mvs)["IMAG"]
    want = np.linalg.lstsq(data.loc[:, mvs].values, Z["IMAG"].values, rcond=None)[0]       # gelsd: minimum norm
    assert_close(got.values, want, 1e-8, 1e-11)
    assert abs(got["imag1"] - got["imag1dup"]) < 1e-12


def test_plspm_objects_hold_no_reference_cycle():
    """A Plspm object (and its Bootstrap) must die with its last reference -- device handle, records in HBM and frames released at that moment, not
    whenever Python's cycle collector next runs (which is inside some later, possibly timed, Plspm() call: the 3-8 ms a bootstrap call seemed to
    cost over a plain one in the round-3 bench were the previous model's garbage being traversed).  Checked with the collector switched off."""
    import gc
    import weakref
    from plspm.plspm import Plspm
    sat, cfg = sat_config("AAAAAA", 1)
    gc.collect()
    gc.disable()
    try:
        for boot in (False, True):
            m = Plspm(sat, cfg, Scheme.CENTROID, bootstrap=boot, bootstrap_iterations=200, processes=1, seed=3)
            m.scores(); m.inner_summary(); m.outer_model(); m.unidimensionality()
            if boot:
                m.bootstrap().weights()
            refs = [weakref.ref(m), weakref.ref(m._result.native)] + ([weakref.ref(m._bootstrap)] if boot else [])
            del m
            assert all(r() is None for r in refs), "a reference cycle keeps the model alive (bootstrap=%s)" % boot
    finally:
        gc.enable()
