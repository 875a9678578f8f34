"""GPU parity of the non-metric missing-data path (plspm_model_set_incomplete_rows, csrc/solver_nmx.h) through the C-ABI and the
host API: against the NaN-aware oracle, the reference-generated golden g13 (fits and bootstrap rows) and the reference's own
expected CSV (tests/test_regression_nonmetric.py:122-138)."""
import os

import numpy as np
import pandas as pd
import pytest

import plspm_oracle as orc
from helpers import GOLDEN, assert_close, load
from test_oracle_golden import RUSSA_C, RUSSA_M_BLOCKS, RUSSA_M_COLS, russa_missing_matrix
from test_solver_hostemu_nmx import split_incomplete

pytestmark = pytest.mark.gpu
SCHEME_ID = {"centroid": 0, "factorial": 1, "path": 2}
RTOL, ATOL = 1e-6, 1e-9


def gpu_model(Xnan, model):
    from plspm import _native
    order = model.mv_order
    P = Xnan.shape[1]
    filled, rows, Mk = split_incomplete(np.ascontiguousarray(Xnan[:, order]))
    boff = np.concatenate(([0], np.cumsum([len(b) for b in model.blocks]))).astype(np.int32)
    modes = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
    nm = _native.NativeModel(boff, model.C.astype(np.uint8), modes, SCHEME_ID[model.scheme], True, model.max_iter, model.tol, 0, nonmetric=True)
    nm.upload(filled)
    if len(rows):
        nm.set_incomplete_rows(rows, Mk > 0, raw_scale=all(k == "RAW" for k in model.scales))
    inv = np.empty(P, dtype=np.int64); inv[order] = np.arange(P)
    return nm, inv


def check_fit(g, r, inv, tag=""):
    assert g["status"] == 0, tag
    assert g["iterations"] == r["iterations"], "%s: iterations %d vs oracle %d" % (tag, g["iterations"], r["iterations"])
    assert_close(g["weights"][inv], r["weights"], RTOL, what=tag + " weights")
    assert_close(g["loadings"][inv], r["loadings"], RTOL, what=tag + " loadings")
    assert_close(g["crossloadings"][inv], r["crossloadings"], RTOL, ATOL)
    assert_close(g["path_coef"], r["path_coef"], RTOL, ATOL)
    assert_close(g["r2"], r["r2"], RTOL, ATOL)
    assert_close(g["total"], r["total"], RTOL, ATOL)
    assert_close(g["scores"], r["scores"], 1e-7, 1e-9, what=tag + " scores")


def rows_in_data_order(rows, inv, P, L, ne):
    return np.concatenate((rows[:, :P][:, inv], rows[:, P:P + L + 2 * ne], rows[:, P + L + 2 * ne:][:, inv]), axis=1)


@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
def test_russa_missing_fit_and_bootstrap_vs_reference_golden(scheme):
    gold = load("g13_nonmetric_missing")
    X = russa_missing_matrix()
    model = orc.Model(RUSSA_M_BLOCKS, RUSSA_C, "AAA", scheme, True, tol=1e-7, scales=["NUM"] * 9)
    nm, inv = gpu_model(X, model)
    g = nm.fit(want_scores=True)
    check_fit(g, orc.fit(X, model), inv, scheme)
    key = "russa_" + scheme
    assert g["iterations"] == int(gold[key + "/iters"])
    assert_close(g["weights"][inv], gold[key + "/weights"], RTOL)
    assert_close(g["loadings"][inv], gold[key + "/loadings"], RTOL)
    assert_close(g["scores"], gold[key + "/scores"], 1e-7, 1e-9)
    rows, status, iters = nm.bootstrap(6, idx=gold["idx47"])
    assert np.all(status == 0) and np.array_equal(iters, gold[key + "/boot_iters"])
    assert_close(rows_in_data_order(rows, inv, 9, 3, nm.n_eff), gold[key + "/boot_rows"], RTOL, 1e-8)


def test_russa_missing_raw_scale_vs_reference_golden():
    gold = load("g13_nonmetric_missing")
    X = russa_missing_matrix()
    model = orc.Model(RUSSA_M_BLOCKS, RUSSA_C, "AAA", "centroid", True, tol=1e-7, scales=["RAW"] * 9)
    nm, inv = gpu_model(X, model)
    g = nm.fit(want_scores=True)
    check_fit(g, orc.fit(X, model), inv, "raw")
    key = "russa_raw_centroid"
    assert g["iterations"] == int(gold[key + "/iters"])
    assert_close(g["weights"][inv], gold[key + "/weights"], RTOL)
    assert_close(g["scores"], gold[key + "/scores"], 1e-7, 1e-9)
    rows, status, iters = nm.bootstrap(3, idx=gold["idx47"][:3])
    assert np.all(status == 0) and np.array_equal(iters, gold[key + "/boot_iters"])
    assert_close(rows_in_data_order(rows, inv, 9, 3, nm.n_eff), gold[key + "/boot_rows"], RTOL, 1e-8)
    # through the host API
    import plspm.config as c
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scale import Scale
    from plspm.scheme import Scheme
    russa = pd.DataFrame(X, columns=RUSSA_M_COLS)
    s = c.Structure(); s.add_path(["AGRI", "IND"], ["POLINS"])
    config = c.Config(s.path(), default_scale=Scale.RAW)
    config.add_lv("AGRI", Mode.A, c.MV("gini"), c.MV("farm"), c.MV("rent"))
    config.add_lv("IND", Mode.A, c.MV("gnpr"), c.MV("labo"))
    config.add_lv("POLINS", Mode.A, c.MV("ecks"), c.MV("death"), c.MV("demo"), c.MV("inst"))
    calc = Plspm(russa, config, Scheme.CENTROID, 100, 0.0000001)
    assert_close(calc.outer_model().loc[RUSSA_M_COLS, "weight"].values, gold[key + "/weights"], RTOL)


@pytest.mark.parametrize("tag", ["A_path", "M_centroid", "A_factorial"])
def test_synthetic_missing_vs_reference_golden(tag):
    gold = load("g13_nonmetric_missing")
    X = gold["synth"]
    modes, scheme = tag.split("_")
    blocks = [np.arange(4 * j, 4 * j + 4) for j in range(6)]
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA" if modes == "A" else "AAAABB", scheme, True, tol=1e-7, scales=["NUM"] * 24)
    nm, inv = gpu_model(X, model)
    g = nm.fit(want_scores=True)
    check_fit(g, orc.fit(X, model), inv, tag)
    key = "synth_" + tag
    assert_close(g["weights"][inv], gold[key + "/weights"], RTOL)
    rows, status, iters = nm.bootstrap(3, idx=gold["idx300"])
    assert np.all(status == 0) and np.array_equal(iters, gold[key + "/boot_iters"])
    assert_close(rows_in_data_order(rows, inv, 24, 6, nm.n_eff), gold[key + "/boot_rows"], RTOL, 1e-8)


def test_larger_problem_device_resampling_spot_checks():
    """5,000 x 60, 3 % of the rows incomplete (K = 150): device-side resampling, counts looked up in the replicate's row list."""
    from plspm import _native
    C = orc.satisfaction_C()
    X, blocks = orc.synth(5000, C, 10, seed=23)
    rs = np.random.RandomState(23)
    Xn = X.copy()
    for row in rs.choice(5000, size=150, replace=False):
        Xn[row, rs.choice(60, size=rs.randint(1, 4), replace=False)] = np.nan
    model = orc.Model(blocks, C, "AAAAAA", "path", True, tol=1e-7, scales=["NUM"] * 60)
    Xn = orc.filter_missing(Xn, model)
    nm, inv = gpu_model(Xn, model)
    g = nm.fit(want_scores=True)
    check_fit(g, orc.fit(Xn, model), inv, "5k")
    rows, status, iters = nm.bootstrap(48, seed=3)
    assert np.all(status == 0)
    rows = rows_in_data_order(rows, inv, 60, 6, nm.n_eff)
    n = Xn.shape[0]
    corr = orc.correction(n)
    for r in (0, 47):
        mine, its = orc.bootstrap_replicate(Xn, model, _native.bootstrap_indices(3, r, n), corr)
        assert its == iters[r]
        assert_close(rows[r], mine, RTOL, ATOL)


def test_mode_b_block_with_missing_cells_is_reported():
    X = russa_missing_matrix()
    model = orc.Model(RUSSA_M_BLOCKS, RUSSA_C, "BAA", "centroid", True, tol=1e-7, scales=["NUM"] * 9)
    nm, inv = gpu_model(X, model)
    assert nm.fit(want_scores=False)["status"] != 0
    # a replicate that does not draw the incomplete AGRI row is fine with Mode B there
    idx = np.tile(np.arange(47), (2, 1)).astype(np.int32)
    idx[1, 0] = 1                                            # row 0 (the AGRI hole) not drawn
    rows, status, _ = nm.bootstrap(2, idx=idx)
    assert status[0] != 0 and status[1] == 0
    mine, _ = orc.bootstrap_replicate(X, model, idx[1], orc.correction(47))
    assert_close(rows_in_data_order(rows, inv, 9, 3, nm.n_eff)[1], mine, RTOL, 1e-8)


def test_api_reproduces_reference_russa_missing_data_test():
    """Mirrors reference tests/test_regression_nonmetric.py:122-138."""
    import plspm.config as c
    import plspm.util as util
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scale import Scale
    from plspm.scheme import Scheme
    ref = os.path.join(GOLDEN, "ref_data")
    russa = pd.read_csv(os.path.join(ref, "russa.csv"), index_col=0)
    russa.iloc[0, 0] = np.nan; russa.iloc[3, 3] = np.nan; russa.iloc[5, 5] = np.nan
    s = c.Structure(); s.add_path(["AGRI", "IND"], ["POLINS"])

    def build(mode_agri=Mode.A):
        config = c.Config(s.path(), default_scale=Scale.NUM)
        config.add_lv("AGRI", mode_agri, c.MV("gini"), c.MV("farm"), c.MV("rent"))
        config.add_lv("IND", Mode.A, c.MV("gnpr"), c.MV("labo"))
        config.add_lv("POLINS", Mode.A, c.MV("ecks"), c.MV("death"), c.MV("demo"), c.MV("inst"))
        return config
    calc = Plspm(russa, build(), Scheme.CENTROID, 100, 0.0000001)
    expected = pd.read_csv(os.path.join(ref, "russa.missing.inner_summary.csv"), index_col=0)
    np.testing.assert_allclose(util.sort_cols(expected.drop(["type"], axis=1)).sort_index(),
                               util.sort_cols(calc.inner_summary().drop(["type", "r_squared_adj"], axis=1)).sort_index().astype(float))
    pd.testing.assert_series_equal(expected.loc[:, "type"].sort_index(), calc.inner_summary().loc[:, "type"].sort_index())
    assert calc.unidimensionality().drop(["mode", "mvs"], axis=1).isnull().values.all()
    gold = load("g13_nonmetric_missing")
    assert_close(calc.scores().loc[:, ["AGRI", "IND", "POLINS"]].values, gold["russa_centroid/scores"], 1e-7, 1e-9)
    assert_close(calc.outer_model().loc[RUSSA_M_COLS, "weight"].values, gold["russa_centroid/weights"], RTOL)
    with pytest.raises(Exception, match="not supported in mode B"):
        Plspm(russa, build(Mode.B), Scheme.CENTROID, 100, 0.0000001)
    boot = Plspm(russa, build(), Scheme.CENTROID, 100, 0.0000001, bootstrap=True, bootstrap_iterations=200, seed=11).bootstrap()
    w = boot.weights()
    assert np.all(np.isfinite(w[["mean", "std.error"]].values)) and int((boot.status() == 0).sum()) >= 190


def test_digit_planes_prepared_before_the_incomplete_rows_are_rebuilt():
    """upload -> plspm_bootstrap_prepare -> plspm_model_set_incomplete_rows is a natural order for a C-ABI caller.  The set call
    zeroes the incomplete rows of the resident matrix, so digit planes (or column statistics) prepared before it describe rows that no
    longer exist: the library must discard them.  Same records as a handle that never prepared early, and as the fp64 route."""
    from plspm import _native
    C = orc.satisfaction_C()
    X, blocks = orc.synth(3000, C, 4, seed=29)
    rs = np.random.RandomState(5)
    Xn = X.copy()
    for row in rs.choice(3000, size=90, replace=False):
        Xn[row, rs.choice(24, size=rs.randint(1, 3), replace=False)] = np.nan
    model = orc.Model(blocks, C, "AAAAAA", "centroid", True, tol=1e-7, scales=["NUM"] * 24)
    Xn = orc.filter_missing(Xn, model)
    ref, _ = gpu_model(Xn, model)
    want = ref.bootstrap(40, seed=11)
    assert ref.get_option("last_gram_path") == 2
    for fixed_planes in (0, 7):              # automatic plane count (statistics in flight) and a fixed one (planes cut by the prepare call itself)
        order = model.mv_order
        filled, rows, Mk = split_incomplete(np.ascontiguousarray(Xn[:, order]))
        boff = np.concatenate(([0], np.cumsum([len(b) for b in model.blocks]))).astype(np.int32)
        nm = _native.NativeModel(boff, model.C.astype(np.uint8), np.zeros(6, dtype=np.int32), SCHEME_ID[model.scheme], True, model.max_iter, model.tol, 0, nonmetric=True)
        nm.upload(filled)
        if fixed_planes:
            nm.set_option("i8_slices", fixed_planes)
        nm.prepare_bootstrap()
        nm.set_incomplete_rows(rows, Mk > 0, raw_scale=False)
        got = nm.bootstrap(40, seed=11)
        assert nm.get_option("last_gram_path") == 2 and np.all(got[1] == 0)
        assert np.array_equal(got[2], want[2])
        if fixed_planes:
            assert_close(got[0], want[0], 1e-10, 1e-12)
        else:
            assert np.array_equal(got[0], want[0])
        nm.set_option("gram_path", 1)
        f64 = nm.bootstrap(40, seed=11)
        assert np.array_equal(f64[2], want[2])
        assert_close(f64[0], want[0], 1e-9, 1e-12)


@pytest.mark.parametrize("route", ["one_launch", "launch_by_launch", "incomplete_rows"])
def test_a_replicate_that_can_never_converge_keeps_the_record_of_max_iter_plus_one_trips(route):
    """Scale.NUM: a column that is constant in a replicate standardises to NaN (the reference iterates its 101 trips on NaN scores and drops the replicate, weights.py:183-186 /
    bootstrap.py:65-66).  The device ends such a problem at the first NaN stop criterion -- it is absorbing -- and must leave the record of the full run: a status != 0 and
    max_iter + 1 iterations, on the one-launch form, the launch-by-launch step (nm_step) and the step for incomplete rows (nmx_step); its neighbours are untouched.
    (Before the last session of round 6 the column's variance was rounding residue of either sign: a finite column of noise with a weight of 1e-9 and PLSPM_OK, or NaN -- solver_core.h
    nm_column_sd.)"""
    from plspm import _native
    C = orc.chain_C(3)
    X, blocks = orc.synth(120, C, 3, seed=44)
    X[:, 0] = 0.0; X[:4, 0] = 1.0                               # a rare indicator: resamples of rows 4 ... 119 only see a constant
    model = orc.Model(blocks, C, "AAA", "path", True, tol=1e-6, scales=["NUM"] * X.shape[1])
    Xn = X.copy()
    if route == "incomplete_rows":
        Xn[7, 4] = np.nan; Xn[30, 8] = np.nan
    nm, inv = gpu_model(Xn, model)
    if route == "launch_by_launch":
        nm.set_option("nm_wave16", 0)
    rs = np.random.RandomState(3)
    idx = np.vstack([np.arange(120), 4 + rs.randint(116, size=120), rs.randint(120, size=120), 4 + rs.randint(116, size=120)]).astype(np.int32)
    rows, status, iters = nm.bootstrap(4, idx=idx)
    assert status[0] == 0 and status[2] == 0 and status[1] != 0 and status[3] != 0, status
    assert iters[1] == model.max_iter + 1 and iters[3] == model.max_iter + 1 and 1 <= iters[0] < 30 and 1 <= iters[2] < 30, iters
    corr = orc.correction(120)
    for b in (0, 2):
        mine, its = orc.bootstrap_replicate(Xn, model, idx[b], corr)
        assert its == iters[b]
        P = X.shape[1]
        ne = (rows.shape[1] - 2 * P - 3) // 2
        got = np.concatenate((rows[b][:P][inv], rows[b][P:P + 3 + 2 * ne], rows[b][P + 3 + 2 * ne:][inv]))
        assert_close(got, mine, RTOL, ATOL)
    for b in (1, 3):
        with pytest.raises((orc.NotConverged, np.linalg.LinAlgError)), np.errstate(all="ignore"):      # (the path scheme's regression meets the NaN scores first)
            orc.bootstrap_replicate(Xn, model, idx[b], corr)
