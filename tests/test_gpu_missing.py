"""GPU parity of the metric missing-data path: mean imputation on the moments
(plspm_model_set_missing + impute_kernel), fit and bootstrap, against the oracle and the reference-generated golden g10
(fits and bootstrap rows on explicit indices -- the reference re-imputes every resampled data set, bootstrap.py:57)."""
import numpy as np
import pandas as pd
import pytest

import plspm_oracle as orc
from helpers import assert_close, load
from test_oracle_golden import case_modes, satisfaction_oracle_inputs
from test_solver_hostemu_missing import aug_matrix

pytestmark = pytest.mark.gpu
SCHEME_ID = {"centroid": 0, "factorial": 1, "path": 2}
RTOL, ATOL = 1e-6, 1e-9


def gpu_model(Xn, model):
    from plspm import _native
    order = model.mv_order
    P = Xn.shape[1]
    Xaug, ind_of = aug_matrix(Xn[:, order])                  # device column order: [filled data | indicators]
    boff = np.concatenate(([0], np.cumsum([len(b) for b in model.blocks]))).astype(np.int32)
    modes = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
    nm = _native.NativeModel(boff, model.C.astype(np.uint8), modes, SCHEME_ID[model.scheme], model.scaled, model.max_iter, model.tol, 0, missing=ind_of)
    nm.upload(Xaug)
    inv = np.empty(P, dtype=np.int64); inv[order] = np.arange(P)
    return nm, inv


def rows_in_data_order(rows, inv, P, L, ne):
    return np.concatenate((rows[:, :P][:, inv], rows[:, P:P + L + 2 * ne], rows[:, P + L + 2 * ne:][:, inv]), axis=1)


@pytest.mark.parametrize("tag", ["A_centroid_1", "M_path_0"])
def test_fit_and_bootstrap_rows_vs_reference_golden(tag):
    g = load("g10_metric_missing")
    _, blocks, _ = satisfaction_oracle_inputs()
    m, scheme, scaled = tag.split("_")
    model = orc.Model(blocks, orc.satisfaction_C(), case_modes(m), scheme, bool(int(scaled)))
    X = orc.filter_missing(g["data"], model)
    nm, inv = gpu_model(X, model)
    out = nm.fit(want_scores=True)
    assert out["status"] == 0 and out["iterations"] == int(g[tag + "/iters"])
    assert_close(out["weights"][inv], g[tag + "/weights"], RTOL)
    assert_close(out["loadings"][inv], g[tag + "/loadings"], RTOL)
    assert_close(out["path_coef"], g[tag + "/path_coef"], RTOL, ATOL)
    assert_close(out["scores"], g[tag + "/scores"], 1e-7, 1e-9)
    rows, status, iters = nm.bootstrap(5, idx=g["idx"])
    assert np.all(status == 0) and np.array_equal(iters, g[tag + "/boot_iters"])
    assert_close(rows_in_data_order(rows, inv, 27, 6, nm.n_eff), g[tag + "/boot_rows"], RTOL, ATOL)


@pytest.mark.parametrize("modes,scheme,scaled", [("AAAAAA", "factorial", False), ("BABABA", "centroid", True)])
def test_synthetic_missing_device_resampling_vs_oracle(modes, scheme, scaled):
    """2,000 x 60 with 3 % of the cells missing in 40 of the columns (P + n_ind = 100 device columns -> T = 8 Gram tiles,
    solver on T = 4): device-side Philox resampling, spot-checked against the oracle on the mirrored indices."""
    from plspm import _native
    C = orc.satisfaction_C()
    X, blocks = orc.synth(2000, C, 10, seed=12)
    rs = np.random.RandomState(12)
    Xn = X.copy()
    cols = rs.choice(60, size=40, replace=False)
    for col in cols:
        Xn[rs.choice(2000, size=60, replace=False), col] = np.nan
    model = orc.Model(blocks, C, modes, scheme, scaled)
    nm, inv = gpu_model(Xn, model)
    out = nm.fit(want_scores=False)
    ref = orc.fit(Xn, model)
    assert out["status"] == 0 and out["iterations"] == ref["iterations"]
    assert_close(out["weights"][inv], ref["weights"], RTOL)
    assert_close(out["loadings"][inv], ref["loadings"], RTOL)
    rows, status, iters = nm.bootstrap(64, seed=21)
    assert np.all(status == 0)
    rows = rows_in_data_order(rows, inv, 60, 6, nm.n_eff)
    corr = orc.correction(2000)
    for r in (0, 31, 63):
        mine, its = orc.bootstrap_replicate(Xn, model, _native.bootstrap_indices(21, r, 2000), corr)
        assert its == iters[r]
        assert_close(rows[r], mine, RTOL, ATOL)


def test_replicate_that_loses_every_present_cell_is_flagged():
    C = orc.satisfaction_C()
    X, blocks = orc.synth(40, C, 3, seed=2)
    Xn = X.copy()
    Xn[:38, 4] = np.nan
    model = orc.Model(blocks, C, "AAAAAA", "centroid", True)
    nm, inv = gpu_model(Xn, model)
    idx = np.tile(np.arange(40), (2, 1)).astype(np.int32)
    idx[1, 38:] = [0, 1]                                     # replicate 1 never draws the two present cells of column 4
    rows, status, iters = nm.bootstrap(2, idx=idx)
    assert status[0] == 0 and status[1] != 0


def test_api_bootstrap_with_missing_values():
    """Plspm(..., bootstrap=True) on data with NaNs: the reference's flow (plspm.py:78-82 -> bootstrap.py:57 -> config.py:300)."""
    import plspm.config as c
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scheme import Scheme
    from helpers import SAT_ADD_ORDER, SAT_PREFIX
    g = load("g10_metric_missing")
    _, blocks, cols = satisfaction_oracle_inputs()
    frame = pd.DataFrame(g["data"], columns=cols)
    structure = c.Structure()
    structure.add_path(["IMAG"], ["EXPE", "SAT", "LOY"]); structure.add_path(["EXPE"], ["QUAL", "VAL", "SAT"])
    structure.add_path(["QUAL"], ["VAL", "SAT"]); structure.add_path(["VAL"], ["SAT"]); structure.add_path(["SAT"], ["LOY"])
    config = c.Config(structure.path(), scaled=True)
    for lv in SAT_ADD_ORDER:
        config.add_lv_with_columns_named(lv, Mode.A, frame, SAT_PREFIX[lv])
    calc = Plspm(frame, config, Scheme.CENTROID, bootstrap=True, bootstrap_iterations=300, seed=4)
    assert calc.iterations() == int(g["A_centroid_1/iters"])
    boot = calc.bootstrap()
    w = boot.weights()
    assert w.shape[0] == 27 and np.all(np.isfinite(w[["mean", "std.error", "perc.025", "perc.975"]].values))
    om = calc.outer_model()
    assert_close(w.loc[om.index, "original"].values, om["weight"].values, 1e-12)
    assert np.all(np.abs(w["mean"] - w["original"]) < 4 * w["std.error"] + 1e-3)
    # replicate 0 of the seeded stream against the oracle
    from plspm import _native
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "centroid", True)
    X = orc.filter_missing(g["data"], model)
    mine, _ = orc.bootstrap_replicate(X, model, _native.bootstrap_indices(4, 0, 249), orc.correction(249))
    rows, status, _ = calc._result.native.bootstrap(1, seed=4)
    assert status[0] == 0
    dev = calc._result.compiled
    inv = np.empty(27, dtype=np.int64); inv[dev.col_index] = np.arange(27)
    assert_close(rows_in_data_order(rows, inv, 27, 6, calc._result.native.n_eff)[0], mine, RTOL, ATOL)
