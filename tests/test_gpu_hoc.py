"""GPU parity of the two-stage higher-order-construct bootstrap (plspm_model_attach_second_stage, csrc/solver_hoc.h) against the
reference-generated golden g12 (mobi, seminr HOC model: fit and bootstrap replicates on explicit indices) and the oracle."""
import os

import numpy as np
import pandas as pd
import pytest

import plspm_oracle as orc
from helpers import GOLDEN, assert_close, load
from test_oracle_golden import MOBI_C1, MOBI_STAGE2, mobi_hoc_inputs, mobi_hoc_model

pytestmark = pytest.mark.gpu
SCHEME_ID = {"centroid": 0, "factorial": 1, "path": 2}
RTOL, ATOL = 1e-6, 1e-9


def handles(tag, max_iter=100):
    from plspm import _native
    g = load("g12_hoc_two_stage")
    X, blocks, _ = mobi_hoc_inputs()
    model1 = mobi_hoc_model(tag, blocks)
    boff1 = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
    modes1 = np.array([0 if m == "A" else 1 for m in model1.modes], dtype=np.int32)
    first = _native.NativeModel(boff1, MOBI_C1.astype(np.uint8), modes1, SCHEME_ID[model1.scheme], True, max_iter, 1e-8, 0, nonmetric=True)
    first.upload(X)
    boff2 = np.array([0, 7, 10, 12, 15, 16], dtype=np.int32)
    modes2 = np.array([modes1[0], 0, 0, 0, 0], dtype=np.int32)
    second = _native.NativeModel(boff2, g[tag + "/path2"].astype(np.uint8), modes2, SCHEME_ID[model1.scheme], True, max_iter, 1e-8, 0, nonmetric=True)
    return first, second, X, model1, g


@pytest.mark.parametrize("tag", ["path_B", "centroid_A"])
def test_two_stage_bootstrap_rows_vs_reference_golden(tag):
    first, second, X, model1, g = handles(tag)
    first.attach_second_stage(second, [0, 1, 2, 4, 5, 6])
    assert first.row_width == 2 * 16 + 5 + 2 * 8
    assert list(first.eff_from) == list(g[tag + "/eff_from"]) and list(first.eff_to) == list(g[tag + "/eff_to"])
    idx = np.vstack([np.arange(250), g["idx"]]).astype(np.int32)
    rows, status, iters = first.bootstrap(5, idx=idx)
    assert np.all(status == 0) and np.array_equal(iters, g[tag + "/iters2"])
    assert_close(rows, g[tag + "/rows"], RTOL, ATOL)
    # device-side resampling + sharding invariance, spot-checked against the oracle
    from plspm import _native
    rows, status, iters = first.bootstrap(96, seed=8)
    assert np.all(status == 0)
    corr = orc.correction(250)
    for r in (0, 95):
        o = orc.fit_two_stage(X[_native.bootstrap_indices(8, r, 250)], model1, MOBI_STAGE2, g[tag + "/path2"], model1.modes[0] + "AAAA", corr)
        assert o["iterations"] == iters[r]
        assert_close(rows[r], np.concatenate((o["weights"], o["r2"], o["total"], o["direct"], o["loadings"])), RTOL, ATOL)
    table, used = first.summary(96, rows[0])                            # on the 96 rows still in HBM
    assert used == 96 and table.shape == (first.row_width, 6)
    assert_close(table[:, 1], rows.mean(axis=0), 1e-10, 1e-12)
    tail, _, _ = first.bootstrap(16, seed=8, rep_offset=80)
    assert np.array_equal(tail, rows[80:])


def test_failed_first_stage_drops_the_replicate():
    first, second, X, model1, g = handles("path_B", max_iter=3)         # stage 1 needs 8+ iterations
    first.attach_second_stage(second, [0, 1, 2, 4, 5, 6])
    rows, status, iters = first.bootstrap(4, seed=1)
    assert np.all(status != 0)


def test_attach_rejects_inconsistent_pairs():
    from plspm._native import NativeBackendError
    first, second, *_ = handles("path_B")
    with pytest.raises(NativeBackendError):
        first.attach_second_stage(second, [0, 1, 2, 3, 5, 6])           # HOC block of 2 columns standing for one 5-column LV
    with pytest.raises(NativeBackendError):
        first.attach_second_stage(second, [0, 1, 2, 4, 5, 5])
    first.attach_second_stage(second, [0, 1, 2, 4, 5, 6])
    with pytest.raises(NativeBackendError):
        first.attach_second_stage(second, [0, 1, 2, 4, 5, 6])           # already paired
    second.close()                                                      # destroy order is free
    first.close()


def test_api_bootstrap_of_a_hoc_model():
    """Plspm(..., bootstrap=True) on the reference's seminr HOC model (tests/test_regression_seminr.py:49-74)."""
    import plspm.config as c
    from plspm import _native
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scale import Scale
    from plspm.scheme import Scheme
    mobi = pd.read_csv(os.path.join(GOLDEN, "ref_data", "mobi.csv"), index_col=0)
    structure = c.Structure()
    structure.add_path(["Expectation", "Quality"], ["Satisfaction"])
    structure.add_path(["Satisfaction"], ["Complaints", "Loyalty"])
    config = c.Config(structure.path(), default_scale=Scale.NUM)
    config.add_higher_order("Satisfaction", Mode.A, ["Image", "Value"])
    config.add_lv_with_columns_named("Expectation", Mode.A, mobi, "CUEX")
    config.add_lv_with_columns_named("Quality", Mode.B, mobi, "PERQ")
    config.add_lv_with_columns_named("Loyalty", Mode.A, mobi, "CUSL")
    config.add_lv_with_columns_named("Image", Mode.A, mobi, "IMAG")
    config.add_lv_with_columns_named("Complaints", Mode.A, mobi, "CUSCO")
    config.add_lv_with_columns_named("Value", Mode.A, mobi, "PERV")
    pls = Plspm(mobi, config, Scheme.PATH, 100, 0.00000001, bootstrap=True, bootstrap_iterations=200, seed=6)
    boot = pls.bootstrap()
    w = boot.weights()
    g = load("g12_hoc_two_stage")
    assert sorted(w.index) == sorted(g["path_B/mvs2"])                  # stage-2 MVs: the constituents appear as MVs of the HOC
    assert np.all(np.isfinite(w[["mean", "std.error", "perc.025", "perc.975"]].values))
    om = pls.outer_model()
    assert_close(w.loc[om.index, "original"].values, om["weight"].values, 1e-12)
    assert_close(w.loc[list(g["path_B/mvs2"]), "original"].values, g["path_B/rows"][0][:16], 1e-6)
    assert np.all(np.abs(w["mean"] - w["original"]) < 5 * w["std.error"] + 1e-3)
    assert boot.r_squared().shape[0] == 3 and boot.total_effects().shape[0] == 8
    assert int((boot.status() == 0).sum()) >= 190
    # replicate 3 of the seeded stream against the oracle
    X, blocks, _ = mobi_hoc_inputs()
    model1 = mobi_hoc_model("path_B", blocks)
    o = orc.fit_two_stage(X[_native.bootstrap_indices(6, 3, 250)], model1, MOBI_STAGE2, g["path_B/path2"], "BAAAA", orc.correction(250))
    mine = boot.replicates()                                             # noqa: SLF001 (rows in device order, failed ones dropped)
    assert boot.status()[:4].tolist() == [0, 0, 0, 0]
    assert_close(mine[3][16:16 + 5 + 16], np.concatenate((o["r2"], o["total"], o["direct"])), RTOL, ATOL)


def _ordinal_hoc(scheme_name):
    import plspm.config as c
    from plspm.mode import Mode
    from plspm.scale import Scale
    from plspm.scheme import Scheme
    mobi = pd.read_csv(os.path.join(GOLDEN, "ref_data", "mobi.csv"), index_col=0).astype(float)
    structure = c.Structure()
    structure.add_path(["Expectation", "Quality"], ["Satisfaction"])
    structure.add_path(["Satisfaction"], ["Complaints", "Loyalty"])
    config = c.Config(structure.path(), default_scale=Scale.ORD)
    config.add_higher_order("Satisfaction", Mode.A, ["Image", "Value"])
    config.add_lv_with_columns_named("Expectation", Mode.A, mobi, "CUEX")
    config.add_lv_with_columns_named("Quality", Mode.A, mobi, "PERQ")
    config.add_lv_with_columns_named("Loyalty", Mode.A, mobi, "CUSL")
    config.add_lv_with_columns_named("Image", Mode.A, mobi, "IMAG")
    config.add_lv_with_columns_named("Complaints", Mode.A, mobi, "CUSCO")
    config.add_lv_with_columns_named("Value", Mode.A, mobi, "PERV")
    return mobi, config, {"path": Scheme.PATH, "centroid": Scheme.CENTROID}[scheme_name]


@pytest.mark.parametrize("tag", ["path", "centroid"])
def test_hoc_on_ordinal_data_fit_and_replicates_vs_reference_golden(tag):
    """Higher order construct on Scale.ORD data (golden g15 from the real reference: full sample + five resamples on explicit indices).
    Round 3: the bootstrap of such a model is batched like every other one -- both stages of every replicate on the device (stage-1
    categorical solver -> stage-2 moments by congruence with the stage-1 score maps over the indicator columns -> stage-2 categorical
    solver; estimator.two_stage_bootstrap_handles) -- and the rows of the explicit index lists must be the reference's
    (estimator.py:43-52 inside bootstrap.py:54-66)."""
    import plspm.weights as w
    from plspm.estimator import Estimator
    from plspm.plspm import Plspm
    g = load("g15_hoc_ordinal")
    mobi, config, scheme = _ordinal_hoc(tag)
    pls = Plspm(mobi, config, scheme, 100, 1e-7)
    fit = pls._result
    cm = fit.compiled
    assert list(cm.dev_mvs) == list(g[tag + "/mvs2"]) and list(cm.lvs) == list(g[tag + "/lvs2"])
    gold = g[tag + "/rows"]
    full = np.concatenate((fit.raw["weights"], fit.raw["r2"], fit.raw["total"], fit.raw["direct"], fit.raw["loadings"]))
    assert_close(full, gold[0], RTOL, ATOL)
    observations = config.filter(mobi)
    calculator = w.WeightsCalculatorFactory(config, 100, 1e-7, np.sqrt(250 / 249), scheme, 0)
    pair = Estimator(config).two_stage_bootstrap_handles(calculator, observations)
    assert list(pair.compiled.dev_mvs) == list(cm.dev_mvs)
    rows, status, iters = pair.native.bootstrap(5, idx=np.ascontiguousarray(g["idx"], dtype=np.int32))
    assert np.all(status == 0) and np.all(iters > 0)
    assert_close(rows, gold[1:], RTOL, ATOL)
    table, used = pair.native.summary(5, full)
    assert used == 5
    assert_close(table[:, 1], gold[1:].mean(axis=0), 1e-6, 1e-9)
    # ... and the device RNG's replicates equal one two-stage device ESTIMATE of the resampled observations each
    from plspm import _native
    rows_d, status_d, _ = pair.native.bootstrap(6, seed=3)
    for r in (0, 5):
        res = Estimator(config).run(calculator, observations.iloc[_native.bootstrap_indices(3, r, 250), :], want_scores=False)
        one = np.concatenate((res.raw["weights"], res.raw["r2"], res.raw["total"], res.raw["direct"], res.raw["loadings"]))
        assert status_d[r] == 0
        assert_close(rows_d[r], one, 1e-7, 1e-10)
    # the stop-rule passes of both stages ran on category codes (the second stage files the first stage's rows under its own blocks);
    # with the multiply-add pass instead ("nm_codes" 0 on both handles) the records are the same bits
    assert pair.native.get_option("last_nm_codes") == 1 and pair.native._second.get_option("last_nm_codes") == 1
    rows_c, status_c, iters_c = pair.native.bootstrap(150, seed=11)
    pair.native.set_option("nm_codes", 0); pair.native._second.set_option("nm_codes", 0)
    rows_m, status_m, iters_m = pair.native.bootstrap(150, seed=11)
    assert pair.native.get_option("last_nm_codes") == 0 and pair.native._second.get_option("last_nm_codes") == 0
    pair.native.set_option("nm_codes", 1); pair.native._second.set_option("nm_codes", 1)
    assert np.array_equal(status_c, status_m) and np.array_equal(iters_c, iters_m)
    assert np.array_equal(rows_c, rows_m, equal_nan=True)


@pytest.mark.parametrize("tag", ["path", "centroid"])
def test_hoc_on_five_point_items_takes_the_matrix_product_pass_in_both_stages(tag):
    """Round 5: with LV blocks of at most 64 indicator columns (mobi recoded to five-point items) the stop-rule passes of BOTH stages of a higher order
    construct run as int8 matrix products (kernels_nmp.h: the second stage files the first stage's rows under its own blocks and reads the composed
    score maps), and the first stage takes the wave step.  Held against the pass on category codes ("nm_mfma" 0 on both handles): same iteration
    counts and statuses, same bits of the records; and a replicate against one two-stage device ESTIMATE of the resampled observations."""
    import plspm.weights as w
    from plspm import _native
    from plspm.estimator import Estimator
    mobi, config, scheme = _ordinal_hoc(tag)
    mobi = np.ceil(mobi / 2.0)
    observations = config.filter(mobi)
    calculator = w.WeightsCalculatorFactory(config, 100, 1e-7, np.sqrt(250 / 249), scheme, 0)
    pair = Estimator(config).two_stage_bootstrap_handles(calculator, observations)
    rows_p, status_p, iters_p = pair.native.bootstrap(200, seed=21)
    assert pair.native.get_option("last_nm_mfma") == 1 and pair.native._second.get_option("last_nm_mfma") == 1
    assert pair.native.get_option("last_nm_wave") == 1
    pair.native.set_option("nm_mfma", 0); pair.native._second.set_option("nm_mfma", 0)
    rows_c, status_c, iters_c = pair.native.bootstrap(200, seed=21)
    assert pair.native.get_option("last_nm_mfma") == 0 and pair.native._second.get_option("last_nm_mfma") == 0
    assert pair.native.get_option("last_nm_codes") == 1 and pair.native._second.get_option("last_nm_codes") == 1
    pair.native.set_option("nm_mfma", 1); pair.native._second.set_option("nm_mfma", 1)
    assert np.array_equal(status_p, status_c) and np.array_equal(iters_p, iters_c)
    assert np.array_equal(rows_p, rows_c, equal_nan=True)
    ok = np.flatnonzero(status_p == 0)
    assert ok.size >= 150
    r = int(ok[-1])
    res = Estimator(config).run(calculator, observations.iloc[_native.bootstrap_indices(21, r, 250), :], want_scores=False)
    one = np.concatenate((res.raw["weights"], res.raw["r2"], res.raw["total"], res.raw["direct"], res.raw["loadings"]))
    assert_close(rows_p[r], one, 1e-7, 1e-10)


def test_api_bootstrap_of_a_hoc_model_on_ordinal_data():
    """Plspm(..., bootstrap=True) on a higher order construct with Scale.ORD data: device RNG index stream, both stages of every
    replicate batched on the device, the reference's frames."""
    from plspm import _native
    from plspm.plspm import Plspm
    mobi, config, scheme = _ordinal_hoc("path")
    pls = Plspm(mobi, config, scheme, 100, 1e-7, bootstrap=True, bootstrap_iterations=40, seed=11)
    boot = pls.bootstrap()
    assert boot.used() >= 38
    w = boot.weights()
    assert np.all(np.isfinite(w[["mean", "std.error", "perc.025", "perc.975"]].values))
    om = pls.outer_model()
    assert_close(w.loc[om.index, "original"].values, om["weight"].values, 1e-12)
    assert np.all(np.abs(w["mean"] - w["original"]) < 6 * w["std.error"] + 1e-3)
    # replicate 0 is the estimate of the device stream's indices (seed, 0)
    observations = config.filter(mobi)
    import plspm.weights as wm
    calculator = wm.WeightsCalculatorFactory(config, 100, 1e-7, np.sqrt(250 / 249), scheme, 0)
    row, _ = Plspm._replicate_runner(config, calculator, observations)(_native.bootstrap_indices(11, 0, 250))
    assert_close(boot.replicates()[0], row, 1e-7, 1e-10)          # (the batched second stage works on moments, the single estimate on the scores)
    assert boot.r_squared().shape[0] == 3 and boot.total_effects().shape[0] == 8


@pytest.mark.parametrize("tol", [1e-7, 1e-30])
def test_long_verification_round_for_the_stragglers_changes_nothing(tol):
    """Round 6 (last session): behind the fourth round of eight steps the first stage's verification takes the stragglers' remaining steps in ONE round (option nm_vlong; the
    list kernel files nothing and the host falls back to short rounds where the slots do not fit -- tol 1e-30: every replicate runs its 101 steps).  Same statuses, iteration
    counts and bits of the records as the short rounds; at 1e-7 the call holds replicates that never converge (first stage: poisoned, PLSPM_NONFINITE after "101" trips of the
    second stage, which a NaN criterion now ends at once)."""
    import plspm.weights as w
    from plspm.estimator import Estimator
    mobi, config, scheme = _ordinal_hoc("path")
    observations = config.filter(mobi)
    calculator = w.WeightsCalculatorFactory(config, 100, tol, np.sqrt(250 / 249), scheme, 0)
    pair = Estimator(config).two_stage_bootstrap_handles(calculator, observations)
    B = 5000 if tol > 1e-10 else 600
    assert pair.native.get_option("nm_vlong") == 1
    a = pair.native.bootstrap(B, seed=1)
    pair.native.set_option("nm_vlong", 0)
    b = pair.native.bootstrap(B, seed=1)
    pair.native.set_option("nm_vlong", 1)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert np.array_equal(a[0], b[0], equal_nan=True)
    bad = a[1] != 0
    if tol > 1e-10:
        assert 1 <= bad.sum() <= 20 and np.all(a[2][bad] == 101) and a[2][~bad].max() < 40
    else:
        assert bad.all()
