"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI, against the oracle,
the golden vectors made from the real reference, and the reference's own R-output CSVs.

Tolerances: north_star demands 1e-6 relative (fp64) on outer weights, LV scores and path coefficients.  The tests
hold the HIP path to 1e-8 relative (+ a 1e-11 absolute floor for quantities that are structurally ~0), i.e. two orders
tighter, and require IDENTICAL iteration counts."""
import numpy as np
import pandas as pd
import pytest

import plspm_oracle as orc
from fuzz_cases import _ragged, _random_dag
from helpers import (GOLDEN, assert_close, case_modes, load, satisfaction_frame, satisfaction_oracle_inputs, SAT_ADD_ORDER,
                     SAT_PREFIX)

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-8, 1e-11
SCHEME_ID = {"centroid": 0, "factorial": 1, "path": 2}


def native_model(model, device_id=0):
    from plspm import _native
    boff = np.concatenate(([0], np.cumsum([len(b) for b in model.blocks]))).astype(np.int32)
    modes = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
    return _native.NativeModel(boff, model.C.astype(np.uint8), modes, SCHEME_ID[model.scheme], model.scaled, model.max_iter, model.tol,
                               device_id)


def gpu_fit(X, model, layout="C"):
    nm = native_model(model)
    Xs = np.asfortranarray(X) if layout == "F" else np.ascontiguousarray(X)
    nm.upload(Xs, model.mv_order.astype(np.int32))
    out = nm.fit(want_scores=True, want_cov=True)
    P = X.shape[1]
    inv = np.empty(P, dtype=np.int64); inv[model.mv_order] = np.arange(P)
    out["weights_d"] = out["weights"][inv]; out["loadings_d"] = out["loadings"][inv]; out["crossloadings_d"] = out["crossloadings"][inv]
    out["pairs"] = list(zip(nm.eff_from.tolist(), nm.eff_to.tolist()))
    out["inv"] = inv
    return nm, out


def check_fit(g, r, tag=""):
    assert g["status"] == 0, tag
    assert g["iterations"] == r["iterations"], "%s: iterations %d vs oracle %d" % (tag, g["iterations"], r["iterations"])
    assert_close(g["weights_d"], r["weights"], RTOL, what=tag + " weights")
    assert_close(g["loadings_d"], r["loadings"], RTOL, what=tag + " loadings")
    assert_close(g["crossloadings_d"], r["crossloadings"], RTOL, ATOL, what=tag + " crossloadings")
    assert_close(g["path_coef"], r["path_coef"], RTOL, ATOL, what=tag + " path coefficients")
    assert_close(g["r2"], r["r2"], RTOL, ATOL, what=tag + " r2")
    assert g["pairs"] == r["effect_pairs"], tag
    assert_close(g["total"], r["total"], RTOL, ATOL); assert_close(g["direct"], r["direct"], RTOL, ATOL)
    assert_close(g["indirect"], r["indirect"], RTOL, ATOL)
    assert_close(g["scores"], r["scores"], 1e-7, 1e-9, what=tag + " scores")
    assert np.array_equal(g["sign"], r["sign"].astype(np.int8)), tag


@pytest.mark.parametrize("modes", ["A", "B", "M"])
@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
@pytest.mark.parametrize("scaled", [False, True])
def test_satisfaction_fit_vs_oracle_and_reference_golden(modes, scheme, scaled):
    X, blocks, cols = satisfaction_oracle_inputs()
    model = orc.Model(blocks, orc.satisfaction_C(), case_modes(modes), scheme, scaled)
    _, g = gpu_fit(X, model)
    check_fit(g, orc.fit(X, model), "%s/%s/%d" % (modes, scheme, scaled))
    gold = load("g1_satisfaction")
    key = "%s_%s_%d" % (modes, scheme, int(scaled))
    assert g["iterations"] == int(gold[key + "/iters"])
    assert_close(g["weights_d"], gold[key + "/weights"], RTOL)
    assert_close(g["path_coef"], gold[key + "/path_coef"], RTOL, ATOL)
    assert_close(g["scores"], gold[key + "/scores"], 1e-7, 1e-9)


def test_column_major_upload_gives_identical_results():
    X, blocks, _ = satisfaction_oracle_inputs()
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "path", True)
    _, a = gpu_fit(X, model, "C")
    _, b = gpu_fit(X, model, "F")
    for k in ("weights", "path_coef", "r2", "scores", "loadings"):
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("n", [250, 1500000 // 6])
def test_scores_into_a_caller_owned_buffer(n):
    """NativeModel.fit(scores_out=...): the C-ABI's contract (the caller owns the [N, L] buffer) without a fresh array per call -- the small block that rides the
    pinned staging area and the large one (> 8 MB) that is copied straight into the buffer; the same bits as the allocating form, wrong shapes refused."""
    C = orc.satisfaction_C()
    X, blocks = orc.synth(n, C, 4, seed=11)
    model = orc.Model(blocks, C, "AAAAAA", "path", True)
    nm = native_model(model)
    nm.upload(X)
    ref = nm.fit(want_scores=True)
    buf = np.full((n, 6), np.nan)
    out = nm.fit(want_scores=True, scores_out=buf)
    assert out["scores"] is buf and np.array_equal(buf, ref["scores"]) and np.array_equal(out["weights"], ref["weights"])
    for bad in (np.empty((n, 5)), np.empty((n, 6), dtype=np.float32), np.empty((6, n)).T):
        with pytest.raises(ValueError):
            nm.fit(want_scores=True, scores_out=bad)


@pytest.mark.parametrize("modes,scheme", [("A", "path"), ("B", "factorial"), ("M", "centroid")])
def test_synth2000_vs_golden(modes, scheme):
    gold = load("g2_synth2000")
    X, blocks = orc.synth(2000, orc.satisfaction_C(), 10, seed=7)
    model = orc.Model(blocks, orc.satisfaction_C(), case_modes(modes, mixed="BABABA"), scheme, True)
    _, g = gpu_fit(X, model)
    key = "%s_%s_1" % (modes, scheme)
    assert g["iterations"] == int(gold[key + "/iters"])
    assert_close(g["weights_d"], gold[key + "/weights"], RTOL)
    assert_close(g["path_coef"], gold[key + "/path_coef"], RTOL, ATOL)
    assert_close(g["r2"], gold[key + "/r2"], RTOL, ATOL)
    assert_close(g["loadings_d"], gold[key + "/loadings"], RTOL)


def test_config2_synth10k_path_single_fit():
    """BASELINE.json configs[1]: 10,000 x 60 x 6, Mode A, Scheme.PATH, single fit."""
    gold = load("g3_synth10k_path")
    X, blocks = orc.synth(10000, orc.satisfaction_C(), 10, seed=0)
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "path", True)
    _, g = gpu_fit(X, model)
    assert g["iterations"] == int(gold["iters"]) == 3
    assert_close(g["weights_d"], gold["weights"], RTOL)
    assert_close(g["path_coef"], gold["path_coef"], RTOL, ATOL)
    assert_close(g["r2"], gold["r2"], RTOL, ATOL)
    assert_close(g["loadings_d"], gold["loadings"], RTOL)
    assert_close(g["scores"][:64], gold["scores_head"], 1e-7, 1e-9)
    assert_close(g["scores"][::97], gold["scores_sample"], 1e-7, 1e-9)
    assert_close(g["scores"].T @ g["scores"], gold["scores_gram"], 1e-7, 1e-7)


@pytest.mark.parametrize("tag", ["B_factorial_1", "A_path_1", "B_centroid_1"])
def test_chain20_wide_gram_vs_golden(tag):
    """Reduced BASELINE.json configs[4] (P=200, L=20; tile-split MFMA Gram, solver with S in global memory)."""
    gold = load("g7_chain20")
    C = orc.chain_C(20)
    X, blocks = orc.synth(4000, C, 10, seed=3)
    m, scheme, _ = tag.split("_")
    model = orc.Model(blocks, C, m * 20, scheme, True)
    _, g = gpu_fit(X, model)
    assert g["status"] == 0 and g["iterations"] == int(gold[tag + "/iters"])
    assert_close(g["weights_d"], gold[tag + "/weights"], RTOL)
    assert_close(g["path_coef"], gold[tag + "/path_coef"], RTOL, ATOL)
    assert_close(g["r2"], gold[tag + "/r2"], RTOL, ATOL)
    assert_close(g["crossloadings_d"], gold[tag + "/crossloadings"], RTOL, ATOL)


@pytest.mark.parametrize("P_per,L", [(3, 2), (5, 7), (10, 7), (9, 11), (13, 9), (13, 10), (12, 14), (10, 20), (13, 17), (15, 15)])
def test_every_gram_tile_count(P_per, L):
    """T = 2, 4, 5, 6, ..., 16 tile configurations of the MFMA Gram: whole 32-column groups (even T) and, for metric models with
    5 <= T <= 15, the odd counts whose last tile has no partner (70 -> T 5, 99 -> 7, 130 -> 9, 117 -> 8, 168 -> 11, 200 -> 13, 225 -> 15)."""
    C = orc.chain_C(L)
    X, blocks = orc.synth(1501, C, P_per, seed=L)
    model = orc.Model(blocks, C, "A" * L, "factorial", True)
    _, g = gpu_fit(X, model)
    check_fit(g, orc.fit(X, model), "P=%d" % (P_per * L))
    Xt = orc.treat_metric(X[:, model.mv_order], True)
    assert_close(g["cov"], Xt.T @ Xt / X.shape[0], 1e-10, 1e-13, what="treated covariance")
    assert_close(g["mean"], X[:, model.mv_order].mean(axis=0), 1e-12, 1e-13)


@pytest.mark.parametrize("P_per,L,modes", [(10, 7, "A"), (10, 20, "B")])
def test_bootstrap_with_odd_tile_counts_vs_oracle(P_per, L, modes):
    """The gathered (bootstrap) walk of gram_wide_kernel with a partner-less last tile (T = 5 and T = 13, the configs[4] geometry)."""
    from plspm import _native
    C = orc.chain_C(L)
    X, blocks = orc.synth(900, C, P_per, seed=40 + L)
    model = orc.Model(blocks, C, modes * L, "factorial", True)
    nm, g = gpu_fit(X, model)
    check_fit(g, orc.fit(X, model), "odd tiles")
    rows, status, iters = nm.bootstrap(6, seed=9)
    assert np.all(status == 0)
    for r in (0, 5):
        mine, its = orc.bootstrap_replicate(X, model, _native.bootstrap_indices(9, r, 900), orc.correction(900))
        assert its == iters[r]
        assert_close(rows[r], mine, RTOL, ATOL, what="replicate %d" % r)


@pytest.mark.parametrize("tag", ["A_centroid_0", "B_path_1", "M_factorial_1"])
def test_bootstrap_explicit_indices_vs_reference_rows(tag):
    """Identical resample indices -> identical replicate rows (reference bootstrap.py:56-64)."""
    gold = load("g4_satisfaction_boot")
    X, blocks, _ = satisfaction_oracle_inputs()
    m, scheme, scaled = tag.split("_")
    model = orc.Model(blocks, orc.satisfaction_C(), case_modes(m), scheme, bool(int(scaled)))
    nm = native_model(model)
    nm.upload(X, model.mv_order.astype(np.int32))
    rows, status, iters = nm.bootstrap(8, idx=gold["idx"])
    assert np.all(status == 0)
    assert np.array_equal(iters, gold[tag + "/iters"])
    P, L, ne = 27, 6, nm.n_eff
    inv = np.empty(P, dtype=np.int64); inv[model.mv_order] = np.arange(P)
    mine = np.concatenate((rows[:, :P][:, inv], rows[:, P:P + L + 2 * ne], rows[:, P + L + 2 * ne:][:, inv]), axis=1)
    assert_close(mine, gold[tag + "/rows"], RTOL, ATOL, what=tag)


def test_bootstrap_10k_seeded_indices_vs_reference_rows():
    gold = load("g3_synth10k_path")
    X, blocks = orc.synth(10000, orc.satisfaction_C(), 10, seed=0)
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "path", True)
    nm = native_model(model)
    nm.upload(X)
    idx = np.stack([np.random.RandomState(int(s)).randint(10000, size=10000) for s in gold["boot_seeds"]]).astype(np.int32)
    rows, status, iters = nm.bootstrap(len(idx), idx=idx)
    assert np.all(status == 0) and np.array_equal(iters, gold["boot_iters"])
    assert_close(rows, gold["boot_rows"], RTOL, ATOL)


def test_device_rng_matches_host_mirror_and_sharding_is_invariant():
    from plspm import _native
    X, blocks = orc.synth(3000, orc.satisfaction_C(), 10, seed=1)
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "centroid", True)
    nm = native_model(model)
    nm.upload(X)
    B, seed = 24, 0xC0FFEE
    rows, status, iters = nm.bootstrap(B, seed=seed)
    idx = np.stack([_native.bootstrap_indices(seed, r, 3000) for r in range(B)])
    assert idx.min() >= 0 and idx.max() < 3000
    rows2, _, iters2 = nm.bootstrap(B, idx=idx)
    assert np.array_equal(rows, rows2) and np.array_equal(iters, iters2)          # bit-identical: same list, same kernels
    a, _, _ = nm.bootstrap(10, seed=seed, rep_offset=0)
    b, _, _ = nm.bootstrap(14, seed=seed, rep_offset=10)
    assert np.array_equal(np.concatenate((a, b)), rows)                            # any sharding reproduces the stream
    corr = orc.correction(3000)
    for r in (0, 7, 23):
        mine, its = orc.bootstrap_replicate(X, model, idx[r], corr)
        assert its == iters[r]
        assert_close(rows[r], mine, RTOL, ATOL)


def test_bootstrap_rejects_out_of_range_index():
    from plspm import _native
    X, blocks, _ = satisfaction_oracle_inputs()
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "centroid", False)
    nm = native_model(model)
    nm.upload(X, model.mv_order.astype(np.int32))
    idx = np.zeros((2, 250), dtype=np.int32); idx[1, 5] = 250
    with pytest.raises(_native.NativeBackendError):
        nm.bootstrap(2, idx=idx)


def test_sign_rule_and_status_codes():
    g5 = load("g5_sign_rule")
    model = orc.Model([np.arange(0, 2), np.arange(2, 7), np.arange(7, 11)], g5["C"], "AAA", "centroid", True)
    _, g = gpu_fit(g5["X"], model)
    check_fit(g, orc.fit(g5["X"], model))
    assert g["sign"][0] == -1 and np.all(g["weights"][:2] > 0)
    X, blocks, _ = satisfaction_oracle_inputs()
    hard = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "centroid", False, max_iter=2, tol=1e-30)
    _, g = gpu_fit(X, hard)
    assert g["status"] == 1 and g["iterations"] == 3
    Xd = X.copy(); Xd[:, blocks[2][1]] = 3.0                      # a constant MV: weight 0, loading 0, cross-loadings NaN -- as the reference returns it (round 6)
    flat = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "centroid", True)
    _, g = gpu_fit(Xd, flat)
    r = orc.fit(Xd, flat)
    assert g["status"] == 0 and g["iterations"] == r["iterations"]
    assert_close(g["weights_d"], r["weights"], RTOL, 1e-12); assert_close(g["loadings_d"], r["loadings"], RTOL, 1e-12)
    assert r["loadings"][blocks[2][1]] == 0.0 and g["loadings_d"][blocks[2][1]] == 0.0
    Xn = X.copy(); Xn[7, 3] = np.nan                              # non-finite data stay flagged
    _, g = gpu_fit(Xn, flat)
    assert g["status"] in (2, 3)


G14_CASES = [("a", "B", "centroid", 1), ("a", "B", "path", 0), ("a", "B", "path", 1), ("a", "M", "factorial", 1), ("a", "M", "path", 0),
             ("b", "A", "path", 1), ("b", "B", "path", 1), ("b", "A", "centroid", 1), ("b", "B", "factorial", 1)]


@pytest.mark.parametrize("which,modes,scheme,scaled", G14_CASES)
def test_rank_deficient_least_squares_match_the_reference_minimum_norm(which, modes, scheme, scaled):
    """Golden g14 (made from the real reference): Mode-B blocks with a duplicated / linearly dependent MV (scipy lstsq = gelsd,
    mode.py:51) and exactly collinear predecessor scores (statsmodels pinv, scheme.py:50, inner_model.py:69).  The device falls
    back from Cholesky to the eigen-truncated pseudo-inverse for the flagged regression only -- fit and bootstrap replicates."""
    from test_oracle_golden import g14_case
    gold = load("g14_rank_deficient")
    X, blocks, C = g14_case(gold, which)
    L = len(blocks)
    model = orc.Model(blocks, C, case_modes(modes, L, mixed="BABABA"), scheme, bool(scaled))
    nm, g = gpu_fit(X, model)
    check_fit(g, orc.fit(X, model), "g14 " + which)
    key = "a_%s_%s_%d" % (modes, scheme, scaled) if which == "a" else "b_%s_%s" % (modes, scheme)
    assert g["iterations"] == int(gold[key + "/iters"])
    assert_close(g["weights_d"], gold[key + "/weights"], RTOL, what=key)
    assert_close(g["path_coef"], gold[key + "/path_coef"], RTOL, ATOL, what=key)
    assert_close(g["scores"], gold[key + "/scores"], 1e-7, 1e-9, what=key)
    if key + "/boot_rows" in gold.files:
        n_idx = len(gold[key + "/boot_rows"])
        rows, status, iters = nm.bootstrap(n_idx, idx=gold["idx"][:n_idx])
        assert np.all(status == 0) and np.array_equal(iters, gold[key + "/boot_iters"])
        assert_close(rows, gold[key + "/boot_rows"], RTOL, 1e-10, what=key + " bootstrap rows")


def test_bootstrap_full_size_properties():
    """BASELINE.json configs[2] at full size (5,000 replicates of 10k x 60 x 6): size-independent properties."""
    X, blocks = orc.synth(10000, orc.satisfaction_C(), 10, seed=0)
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "path", True)
    nm = native_model(model)
    nm.upload(X)
    rows, status, iters = nm.bootstrap(5000, seed=1)
    assert np.all(status == 0) and iters.min() >= 2 and iters.max() <= 6
    P, L, ne = 60, 6, nm.n_eff
    w, r2, tot, direct, ld = rows[:, :P], rows[:, P:P + L], rows[:, P + L:P + L + ne], rows[:, P + L + ne:P + L + 2 * ne], rows[:, -P:]
    assert np.all(np.isfinite(rows))
    assert np.all(r2[:, 0] == 0) and np.all((r2[:, 1:] > 0) & (r2[:, 1:] < 1))
    assert np.all(np.abs(ld) <= 1 + 1e-12) and np.all(ld > 0) and np.all(w > 0)
    base = orc.fit(X, model)
    # bootstrap means sit on the full-sample estimates (bias << spread), spread ~ 1/sqrt(N)
    assert np.max(np.abs(w.mean(axis=0) - base["weights"])) < 5e-4
    assert np.max(np.abs(direct.mean(axis=0) - base["direct"])) < 3e-3
    assert 0.002 < w.std(axis=0).max() < 0.02
    # total = direct + indirect >= structure: pairs without a direct edge have direct == 0 exactly
    pairs = list(zip(nm.eff_from.tolist(), nm.eff_to.tolist()))
    Cm = orc.satisfaction_C()
    for e, (f, t) in enumerate(pairs):
        if Cm[t, f] == 0:
            assert np.all(direct[:, e] == 0) and np.all(tot[:, e] != 0)
    # idempotence: the same seed reproduces the batch bit for bit
    again, _, _ = nm.bootstrap(5000, seed=1)
    assert np.array_equal(rows, again)


def test_plspm_api_reproduces_reference_test_satisfaction():
    """Mirror of reference tests/test_regression_metric.py:32-94 through the drop-in API."""
    import math
    import plspm.config as c
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scheme import Scheme
    import plspm.util as util
    import os
    sat = satisfaction_frame()
    structure = c.Structure()
    structure.add_path(["IMAG"], ["EXPE", "SAT", "LOY"]); structure.add_path(["EXPE"], ["QUAL", "VAL", "SAT"])
    structure.add_path(["QUAL"], ["VAL", "SAT"]); structure.add_path(["VAL"], ["SAT"]); structure.add_path(["SAT"], ["LOY"])
    config = c.Config(structure.path(), scaled=False)
    for lv in SAT_ADD_ORDER:
        config.add_lv_with_columns_named(lv, Mode.A, sat, SAT_PREFIX[lv])
    ref = os.path.join(GOLDEN, "ref_data")
    calc = Plspm(sat, config)
    expected_scores = pd.read_csv(os.path.join(ref, "satisfaction.scores.csv"))
    np.testing.assert_allclose(util.sort_cols(expected_scores), util.sort_cols(calc.scores()))
    inner = pd.read_csv(os.path.join(ref, "satisfaction.inner-model.csv"), index_col=0)
    actual = calc.inner_model()
    actual = actual[actual["to"].isin(["SAT"])].drop(["to"], axis=1)
    np.testing.assert_allclose(util.sort_cols(inner).sort_index(), util.sort_cols(actual.set_index(["from"], drop=True)).sort_index())
    outer = pd.read_csv(os.path.join(ref, "satisfaction.outer-model.csv"), index_col=0).drop(["block"], axis=1)
    pd.testing.assert_index_equal(outer.columns, calc.outer_model().columns)
    np.testing.assert_allclose(util.sort_cols(outer.sort_index()), util.sort_cols(calc.outer_model()).sort_index())
    cl = pd.read_csv(os.path.join(ref, "satisfaction.crossloadings.csv"), index_col=0)
    np.testing.assert_allclose(util.sort_cols(cl.drop(["block"], axis=1)).sort_index(), util.sort_cols(calc.crossloadings()).sort_index())
    summ = pd.read_csv(os.path.join(ref, "satisfaction.inner-summary.csv"), index_col=0)
    np.testing.assert_allclose(util.sort_cols(summ.drop(["type"], axis=1)).sort_index(),
                               util.sort_cols(calc.inner_summary().drop(["type", "r_squared_adj"], axis=1)).sort_index())
    pd.testing.assert_series_equal(summ.loc[:, "type"].sort_index(), calc.inner_summary().loc[:, "type"].sort_index())
    eff = pd.read_csv(os.path.join(ref, "satisfaction.effects.csv"), index_col=0)
    pd.testing.assert_frame_equal(eff.loc[:, ["from", "to"]].sort_index(), calc.effects().loc[:, ["from", "to"]].sort_index())
    np.testing.assert_allclose(eff.drop(["from", "to"], axis=1).sort_index(), calc.effects().drop(["from", "to"], axis=1).sort_index(),
                               atol=1e-12)
    unidim = pd.read_csv(os.path.join(ref, "satisfaction_unidim.csv"), index_col=0)
    np.testing.assert_allclose(util.sort_cols(unidim.drop(["mode"], axis=1)).sort_index(),
                               util.sort_cols(calc.unidimensionality().drop(["mode"], axis=1)).sort_index().astype(float))
    assert math.isclose(0.609741624338411, calc.goodness_of_fit())
    for scheme, fname in ((Scheme.PATH, "satisfaction.outer-model-path.csv"), (Scheme.FACTORIAL, "satisfaction.outer-model-factorial.csv")):
        exp = util.sort_cols(pd.read_csv(os.path.join(ref, fname), index_col=0).drop(["block"], axis=1)).sort_index()
        np.testing.assert_allclose(exp, util.sort_cols(Plspm(sat, config, scheme).outer_model()).sort_index())


def test_plspm_api_bootstrap_matches_reference_statistical_test():
    """Mirror of reference tests/test_regression_bootstrap.py:20-48 (absolute tolerances 0.05-0.15 as there)."""
    import os
    import plspm.config as c
    import plspm.util as util
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    sat = satisfaction_frame()
    structure = c.Structure()
    structure.add_path(["IMAG"], ["EXPE", "SAT", "LOY"]); structure.add_path(["EXPE"], ["QUAL", "VAL", "SAT"])
    structure.add_path(["QUAL"], ["VAL", "SAT"]); structure.add_path(["VAL"], ["SAT"]); structure.add_path(["SAT"], ["LOY"])
    config = c.Config(structure.path(), scaled=False)
    for lv in ["IMAG", "EXPE", "QUAL", "VAL", "SAT", "LOY"]:
        config.add_lv_with_columns_named(lv, Mode.A, sat, SAT_PREFIX[lv])
    calc = Plspm(sat, config, bootstrap=True, processes=4, seed=7)
    ref = os.path.join(GOLDEN, "ref_data")
    drop = ["t stat."]
    for fname, frame, atol in (("satisfaction_boot_weights.csv", calc.bootstrap().weights(), 0.05),
                               ("satisfaction_boot_rsquared.csv", calc.bootstrap().r_squared(), 0.1),
                               ("satisfaction_boot_total_effects.csv", calc.bootstrap().total_effects(), 0.1),
                               ("satisfaction_boot_paths.csv", calc.bootstrap().paths(), 0.1),
                               ("satisfaction_boot_loadings.csv", calc.bootstrap().loading(), 0.15)):
        expected = pd.read_csv(os.path.join(ref, fname), index_col=0)
        np.testing.assert_allclose(util.sort_cols(expected), util.sort_cols(frame.drop(columns=drop)), atol=atol, err_msg=fname)


def test_config5_full_size_1m_rows_properties():
    """BASELINE.json configs[4] at full size (1,000,000 x 200 x 20, Mode B, FACTORIAL): size-independent properties.
    The N x L scores come from the streaming scores kernel, lv_cov / path coefficients from the Gram + LDS solver:
    two independent device paths that must agree; a 50k-row prefix is checked against the oracle outright."""
    C = orc.chain_C(20)
    X, blocks = orc.synth(1000000, C, 10, seed=0)
    model = orc.Model(blocks, C, "B" * 20, "factorial", True)
    nm, g = gpu_fit(X, model)
    n = X.shape[0]
    assert g["status"] == 0 and 2 <= g["iterations"] <= 4
    s = g["scores"]
    assert np.all(np.isfinite(s))
    assert np.max(np.abs(s.mean(axis=0))) < 1e-9                       # scores are centred
    cov = s.T @ s / n
    assert_close(np.diag(cov), np.ones(20), 1e-9)                      # ... and have unit population variance (weights.py:57-60)
    assert_close(cov, g["lv_cov"], 1e-8, 1e-10, what="streamed scores vs Gram-side LV covariance")
    # inner-model normal equations hold on the streamed scores
    for i in range(20):
        f = np.flatnonzero(C[i])
        if f.size:
            beta = np.linalg.solve(cov[np.ix_(f, f)], cov[f, i])
            assert_close(g["path_coef"][i, f], beta, 1e-7, 1e-10)
    # Mode B optimality: X_b' (z_l - X_b w_l) = 0  <=>  within a block, cov(x_p, score_l) is proportional to (S_bb w)_p
    Xt = X[::20, :]                                                    # 50k-row slice for a cheap loadings check
    sub = s[::20, :]
    ld = np.array([np.corrcoef(Xt[:, p], sub[:, p // 10])[0, 1] for p in range(0, 200, 7)])
    assert_close(g["loadings"][0:200:7], ld, 0, 0.02)                  # statistical: slice vs full sample
    # prefix vs oracle (exact parity on the same 50k rows)
    Xs = np.ascontiguousarray(X[:50000])
    _, gs = gpu_fit(Xs, model)
    check_fit(gs, orc.fit(Xs, model), "50k prefix")


@pytest.mark.parametrize("n", [50000, 70000])
def test_bootstrap_large_n_histogram_paths(n):
    """N = 50,000: the largest LDS histograms (16-bit counters, 100 KB; a count of N must still fit); N = 70,000 > 65,535: the
    global-scratch resampler.  Both must give the same rows as the oracle run on the same indices (device RNG mirrored on the
    host), and explicit indices must reproduce the RNG path bit for bit."""
    from plspm import _native
    C = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0]])
    X, blocks = orc.synth(n, C, 4, seed=11)
    model = orc.Model(blocks, C, "ABA", "path", True)
    nm = native_model(model)
    nm.upload(X)
    rows, status, iters = nm.bootstrap(6, seed=99, rep_offset=3)
    assert np.all(status == 0)
    idx = np.stack([_native.bootstrap_indices(99, 3 + r, n) for r in range(6)])
    rows2, _, iters2 = nm.bootstrap(6, idx=idx)
    assert np.array_equal(rows, rows2) and np.array_equal(iters, iters2)
    corr = orc.correction(n)
    for r in (0, 5):
        mine, its = orc.bootstrap_replicate(X, model, idx[r], corr)
        assert its == iters[r]
        assert_close(rows[r], mine, RTOL, ATOL)
    worst = np.zeros((1, n), dtype=np.int32)              # every draw hits the same row: multiplicity N, zero variance
    _, st, _ = nm.bootstrap(1, idx=worst)
    assert st[0] != 0


# ------------------------------------------------------------------ edge cases (ragged blocks, limits, tiny inputs)
@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
def test_ragged_blocks_including_single_item_constructs(scheme):
    rng = np.random.default_rng(4)
    C = _random_dag(7, rng)
    X, blocks = _ragged(400, C, [1, 3, 1, 7, 2, 12, 1], seed=8)
    for modes in ("AAAAAAA", "ABBABBA"):
        model = orc.Model(blocks, C, modes, scheme, True)
        _, g = gpu_fit(X, model)
        check_fit(g, orc.fit(X, model), "ragged %s %s" % (modes, scheme))


def test_two_lv_model_effects_special_case_and_tiny_n():
    """L == 2 takes the `total = path` branch of inner_model.py:37-38; N = 10 is the reference's bootstrap minimum."""
    C = np.array([[0, 0], [1, 0]])
    X, blocks = _ragged(10, C, [3, 2], seed=2)
    model = orc.Model(blocks, C, "AB", "path", False)
    nm, g = gpu_fit(X, model)
    r = orc.fit(X, model)
    check_fit(g, r, "L=2 N=10")
    assert g["pairs"] == [(0, 1)] and g["indirect"][0] == 0.0
    rng = np.random.RandomState(3)
    idx = rng.randint(10, size=(16, 10)).astype(np.int32)
    rows, status, iters = nm.bootstrap(16, idx=idx)
    corr = orc.correction(10)
    for b in range(16):
        try:
            mine, its = orc.bootstrap_replicate(X, model, idx[b], corr)
        except Exception:
            assert status[b] != 0
            continue
        if status[b] == 0 and np.all(np.isfinite(mine)):
            assert its == iters[b]
            assert_close(rows[b], mine, 1e-7, 1e-9)


def test_limits_p254_and_l64():
    """Largest supported shapes: P = 254 (T = 16 tiles, two workgroups per k-group walk) and L = 64 single-digit blocks."""
    rng = np.random.default_rng(5)
    C = orc.chain_C(64)
    sizes = [4] * 62 + [3, 3]                     # P = 254
    X, blocks = _ragged(1200, C, sizes, seed=6)
    model = orc.Model(blocks, C, "A" * 64, "factorial", True)
    _, g = gpu_fit(X, model)
    check_fit(g, orc.fit(X, model), "P=254 L=64")
    Cd = _random_dag(12, rng, density=0.9)         # dense DAG: up to 11 predecessors -> LDS-scratch Cholesky path (k > 8)
    X2, b2 = _ragged(600, Cd, [3] * 12, seed=7)
    for scheme in ("path", "centroid"):
        m2 = orc.Model(b2, Cd, "ABABABABABAB", scheme, True)
        _, g2 = gpu_fit(X2, m2)
        check_fit(g2, orc.fit(X2, m2), "dense DAG " + scheme)


def test_handle_reuse_and_reupload():
    X, blocks, _ = satisfaction_oracle_inputs()
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "centroid", False)
    nm = native_model(model)
    nm.upload(X, model.mv_order.astype(np.int32))
    a = nm.fit()
    rows1, _, _ = nm.bootstrap(5, seed=2)
    Xh = np.ascontiguousarray(X[:125])
    nm.upload(Xh, model.mv_order.astype(np.int32))                  # new data on the same handle
    b = nm.fit()
    assert b["scores"].shape == (125, 6) and not np.allclose(a["weights"], b["weights"])
    assert_close(b["weights"][np.argsort(np.argsort(model.mv_order))] if False else b["weights"], native_fit_weights(Xh, model), 1e-12)
    nm.upload(X, model.mv_order.astype(np.int32))
    c2 = nm.fit()
    rows2, _, _ = nm.bootstrap(5, seed=2)
    assert np.array_equal(a["weights"], c2["weights"]) and np.array_equal(rows1, rows2)


def native_fit_weights(X, model):
    nm = native_model(model)
    nm.upload(X, model.mv_order.astype(np.int32))
    return nm.fit()["weights"]


def test_api_single_item_constructs_gof_raises():
    """reference tests/test_regression_metric.py:113-125."""
    import plspm.config as c
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scheme import Scheme
    sat = satisfaction_frame()
    structure = c.Structure()
    structure.add_path(["IMAG"], ["EXPE", "SAT", "LOY"]); structure.add_path(["EXPE"], ["QUAL", "VAL", "SAT"])
    structure.add_path(["QUAL"], ["VAL", "SAT"]); structure.add_path(["VAL"], ["SAT"]); structure.add_path(["SAT"], ["LOY"])
    config = c.Config(structure.path())
    for lv in ["QUAL", "VAL", "SAT", "LOY", "IMAG", "EXPE"]:
        config.add_lv(lv, Mode.A, c.MV(SAT_PREFIX[lv] + "1"))
    calc = Plspm(sat, config, Scheme.CENTROID)
    with pytest.raises(ValueError):
        calc.goodness_of_fit()
    assert calc.scores().shape == (250, 6)


def test_api_mode_b_inner_summary_matches_reference_csv():
    """reference tests/test_regression_metric.py:96-111."""
    import os
    import plspm.config as c
    import plspm.util as util
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scheme import Scheme
    sat = satisfaction_frame()
    structure = c.Structure()
    structure.add_path(["IMAG"], ["EXPE", "SAT", "LOY"]); structure.add_path(["EXPE"], ["QUAL", "VAL", "SAT"])
    structure.add_path(["QUAL"], ["VAL", "SAT"]); structure.add_path(["VAL"], ["SAT"]); structure.add_path(["SAT"], ["LOY"])
    config = c.Config(structure.path(), scaled=False)
    for lv in ["QUAL", "VAL", "SAT", "LOY", "IMAG", "EXPE"]:
        config.add_lv_with_columns_named(lv, Mode.B, sat, SAT_PREFIX[lv])
    calc = Plspm(sat, config, Scheme.CENTROID)
    exp = pd.read_csv(os.path.join(GOLDEN, "ref_data", "satisfaction.modeb.inner-summary.csv"), index_col=0)
    np.testing.assert_allclose(util.sort_cols(exp.drop(["type"], axis=1)).sort_index(),
                               util.sort_cols(calc.inner_summary().drop(["type", "r_squared_adj"], axis=1)).sort_index().astype(float))
    pd.testing.assert_series_equal(exp.loc[:, "type"].sort_index(), calc.inner_summary().loc[:, "type"].sort_index())


# ------------------------------------------------------------------ non-metric NUM / RAW
from test_oracle_golden import RUSSA_BLOCKS, RUSSA_C, RUSSA_COLS, russa_inputs     # noqa: E402


def gpu_fit_nm(X, model):
    from plspm import _native
    boff = np.concatenate(([0], np.cumsum([len(b) for b in model.blocks]))).astype(np.int32)
    modes = np.array([0 if m == "A" else 1 for m in model.modes], dtype=np.int32)
    nm = _native.NativeModel(boff, model.C.astype(np.uint8), modes, SCHEME_ID[model.scheme], True, model.max_iter, model.tol, 0, nonmetric=True)
    nm.upload(np.ascontiguousarray(X), model.mv_order.astype(np.int32))
    out = nm.fit(want_scores=True, want_cov=True)
    P = X.shape[1]
    inv = np.empty(P, dtype=np.int64); inv[model.mv_order] = np.arange(P)
    out["weights_d"] = out["weights"][inv]; out["loadings_d"] = out["loadings"][inv]; out["crossloadings_d"] = out["crossloadings"][inv]
    out["pairs"] = list(zip(nm.eff_from.tolist(), nm.eff_to.tolist()))
    return nm, out


def check_fit_nm(g, r, tag=""):
    assert g["status"] == 0, tag
    assert g["iterations"] == r["iterations"], "%s: iterations %d vs oracle %d" % (tag, g["iterations"], r["iterations"])
    assert_close(g["weights_d"], r["weights"], RTOL, what=tag + " weights")
    assert_close(g["loadings_d"], r["loadings"], RTOL, what=tag + " loadings")
    assert_close(g["crossloadings_d"], r["crossloadings"], RTOL, ATOL)
    assert_close(g["path_coef"], r["path_coef"], RTOL, ATOL)
    assert_close(g["r2"], r["r2"], RTOL, ATOL)
    assert g["pairs"] == r["effect_pairs"]
    assert_close(g["total"], r["total"], RTOL, ATOL)
    assert_close(g["scores"], r["scores"], 1e-7, 1e-9, what=tag + " scores")
    assert np.all(g["sign"] == 1)


@pytest.mark.parametrize("modes", ["AAA", "BBB", "ABA"])
@pytest.mark.parametrize("scheme", ["centroid", "factorial", "path"])
def test_nonmetric_russa_vs_oracle_and_reference_golden(modes, scheme):
    X = russa_inputs()
    model = orc.Model(RUSSA_BLOCKS, RUSSA_C, modes, scheme, True, tol=1e-7, scales=["NUM"] * 9)
    _, g = gpu_fit_nm(X, model)
    check_fit_nm(g, orc.fit(X, model), modes + "/" + scheme)
    gold = load("g8_nonmetric_russa")
    for kind in ("NUM", "RAW", "MIX"):                              # RAW / mixed configurations give the same numbers
        key = "%s_%s_%s" % (modes, scheme, kind)
        assert g["iterations"] == int(gold[key + "/iters"])
        assert_close(g["weights_d"], gold[key + "/weights"], RTOL)
        assert_close(g["scores"], gold[key + "/scores"], 1e-7, 1e-9)


@pytest.mark.parametrize("modes,scheme", [("A", "path"), ("B", "factorial"), ("M", "centroid")])
def test_nonmetric_synth2000_vs_golden(modes, scheme):
    gold = load("g9_nonmetric_synth2000")
    X, blocks = orc.synth(2000, orc.satisfaction_C(), 10, seed=7)
    model = orc.Model(blocks, orc.satisfaction_C(), case_modes(modes, mixed="BABABA"), scheme, True, tol=1e-7, scales=["NUM"] * 60)
    _, g = gpu_fit_nm(X, model)
    key = "%s_%s" % (modes, scheme)
    assert g["status"] == 0 and g["iterations"] == int(gold[key + "/iters"])
    assert_close(g["weights_d"], gold[key + "/weights"], RTOL)
    assert_close(g["path_coef"], gold[key + "/path_coef"], RTOL, ATOL)
    assert_close(g["loadings_d"], gold[key + "/loadings"], RTOL)


@pytest.mark.parametrize("tag", ["AAA_centroid_NUM", "ABA_path_NUM"])
def test_nonmetric_bootstrap_explicit_indices_vs_reference_rows(tag):
    gold = load("g8_nonmetric_russa")
    X = russa_inputs()
    modes, scheme, _ = tag.split("_")
    model = orc.Model(RUSSA_BLOCKS, RUSSA_C, modes, scheme, True, tol=1e-7, scales=["NUM"] * 9)
    nm, _ = gpu_fit_nm(X, model)
    rows, status, iters = nm.bootstrap(6, idx=gold["idx"])
    assert np.all(status == 0) and np.array_equal(iters, gold[tag + "/boot_iters"])
    P, L, ne = 9, 3, nm.n_eff
    inv = np.empty(P, dtype=np.int64); inv[model.mv_order] = np.arange(P)
    mine = np.concatenate((rows[:, :P][:, inv], rows[:, P:P + L + 2 * ne], rows[:, P + L + 2 * ne:][:, inv]), axis=1)
    assert_close(mine, gold[tag + "/boot_rows"], RTOL, ATOL)


def test_nonmetric_dense_and_gathering_stop_rule_passes_agree():
    """The bootstrap's dense stop-rule pass (nm_conv_dense_kernel) and the gathering pass it replaced (still used for N > 65,535
    or models too wide even for block staging; option conv_pass = 1 forces it) must take the same decisions and give the same rows."""
    X, blocks = orc.synth(3000, orc.satisfaction_C(), 5, seed=17)
    model = orc.Model(blocks, orc.satisfaction_C(), "ABABAB", "factorial", True, tol=1e-7, scales=["NUM"] * 30)
    nm, _ = gpu_fit_nm(X, model)
    wave = nm.bootstrap(130, seed=2)                        # round 6: one solver launch + verification (tests/test_gpu_nmwave.py); everything below: the per-iteration launches
    assert nm.get_option("last_nm_wave16") == 1
    nm.set_option("nm_wave16", 0)
    dense = nm.bootstrap(130, seed=2)
    assert nm.get_option("last_nm_wave16") == 0
    assert np.array_equal(wave[1], dense[1]) and np.array_equal(wave[2], dense[2])
    assert_close(wave[0], dense[0], 1e-9, 1e-12)
    nm.set_option("conv_pass", 1)                           # the gathering pass
    gathered = nm.bootstrap(130, seed=2)
    nm.set_option("conv_pass", 2)                           # coefficient tile staged one LV block at a time (wide models)
    blocked = nm.bootstrap(130, seed=2)
    nm.set_option("conv_pass", 0)
    for other in (gathered, blocked):
        assert np.array_equal(dense[1], other[1]) and np.array_equal(dense[2], other[2])
        assert_close(dense[0], other[0], 1e-12, 1e-14)
    assert np.all(dense[1] == 0) and dense[2].min() >= 2
    # round 3: the dense pass reads the row multiplicities from the int8 counts of the digit-plane Gram (no second resample kernel, no
    # uint16 histograms, no (row,count) lists); nm_counts8 = 0 is the round-2 route.  Same weights, same sums: bit-identical records,
    # whole and block-staged coefficient tiles alike; and a FIT after a bootstrap must not pick up the last replicate's counts.
    assert nm.get_option("nm_counts8") == 1 and nm.get_option("last_gram_path") == 2
    nm.set_option("nm_counts8", 0)
    old_route = nm.bootstrap(130, seed=2)
    nm.set_option("conv_pass", 2)
    old_blocked = nm.bootstrap(130, seed=2)
    nm.set_option("conv_pass", 0); nm.set_option("nm_counts8", 1)
    for a, b in ((dense, old_route), (blocked, old_blocked)):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    again = nm.fit(want_scores=False)
    assert again["status"] == 0 and again["iterations"] == _["iterations"] and np.array_equal(again["weights"], _["weights"])


def test_nonmetric_live_problem_list_across_batches():
    """The stop-rule pass walks a compacted list of the problems still iterating (active_list_kernel: trips of 1,024 problems).  Replicates
    stop after different numbers of iterations (tight tolerance, 600 rows), so the list shrinks from pass to pass; the records of a
    replicate must not depend on who else is in its batch: 2,300 in one call == the same replicate ids in calls of 100 and of 1."""
    X, blocks = orc.synth(600, orc.satisfaction_C(), 4, seed=23)
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "centroid", True, tol=1e-9, scales=["NUM"] * 24)
    nm, _ = gpu_fit_nm(X, model)
    rows, status, iters = nm.bootstrap(2300, seed=4)
    assert np.all(status == 0) and iters.max() > iters.min()               # a spread of iteration counts: the list shrinks between passes
    for first in (0, 1024, 2200):
        r2, s2, i2 = nm.bootstrap(100, seed=4, rep_offset=first)
        assert np.array_equal(r2, rows[first:first + 100]) and np.array_equal(i2, iters[first:first + 100])
    r1, _, i1 = nm.bootstrap(1, seed=4, rep_offset=2299)
    assert np.array_equal(r1[0], rows[2299]) and i1[0] == iters[2299]
    from plspm import _native
    mine, its = orc.bootstrap_replicate(X, model, _native.bootstrap_indices(4, 1500, 600), orc.correction(600))
    assert its == iters[1500]
    assert_close(rows[1500], mine, RTOL, ATOL)


def test_nonmetric_bootstrap_10k_vs_oracle_spot_checks():
    from plspm import _native
    X, blocks = orc.synth(10000, orc.satisfaction_C(), 10, seed=0)
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "path", True, tol=1e-7, scales=["NUM"] * 60)
    nm, g = gpu_fit_nm(X, model)
    check_fit_nm(g, orc.fit(X, model), "10k nonmetric")
    rows, status, iters = nm.bootstrap(300, seed=5)
    assert np.all(status == 0)
    corr = orc.correction(10000)
    for r in (0, 151, 299):
        mine, its = orc.bootstrap_replicate(X, model, _native.bootstrap_indices(5, r, 10000), corr)
        assert its == iters[r]
        assert_close(rows[r], mine, RTOL, ATOL)


def test_api_nonmetric_reproduces_reference_russa_and_seminr_tests():
    """Mirrors reference tests/test_regression_nonmetric.py:18-63 (russa, Scale.NUM) and tests/test_regression_seminr.py:10-33."""
    import math
    import os
    import plspm.config as c
    import plspm.util as util
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scale import Scale
    from plspm.scheme import Scheme
    ref = os.path.join(GOLDEN, "ref_data")
    russa = pd.read_csv(os.path.join(ref, "russa.csv"), index_col=0)
    s = c.Structure(); s.add_path(["AGRI", "IND"], ["POLINS"])
    config = c.Config(s.path(), default_scale=Scale.NUM)
    config.add_lv("POLINS", Mode.A, c.MV("ecks"), c.MV("death"), c.MV("demo"), c.MV("inst"))
    config.add_lv("AGRI", Mode.A, c.MV("gini"), c.MV("rent"), c.MV("farm"))
    config.add_lv("IND", Mode.A, c.MV("gnpr"), c.MV("labo"))
    calc = Plspm(russa, config, Scheme.CENTROID, 100, 0.0000001)
    np.testing.assert_allclose(util.sort_cols(pd.read_csv(os.path.join(ref, "russa.scores.csv"), index_col=0)), util.sort_cols(calc.scores()))
    inner = pd.read_csv(os.path.join(ref, "russa.inner_model.csv"), index_col=0)
    actual = calc.inner_model()
    actual = actual[actual["to"].isin(["POLINS"])].drop(["to"], axis=1)
    np.testing.assert_allclose(util.sort_cols(inner).sort_index(), util.sort_cols(actual.set_index(["from"], drop=True)).sort_index())
    om = pd.read_csv(os.path.join(ref, "russa.outer_model.csv"), index_col=0)
    np.testing.assert_allclose(util.sort_cols(om.filter(["weight", "loading", "communality", "redundancy"])).sort_index(),
                               util.sort_cols(calc.outer_model()).sort_index())
    cl = pd.read_csv(os.path.join(ref, "russa.crossloadings.csv"), index_col=0)
    np.testing.assert_allclose(util.sort_cols(cl.filter(["AGRI", "IND", "POLINS"])).sort_index(), util.sort_cols(calc.crossloadings()).sort_index())
    summ = pd.read_csv(os.path.join(ref, "russa.inner_summary.csv"), index_col=0)
    np.testing.assert_allclose(util.sort_cols(summ.drop(["type"], axis=1)).sort_index(),
                               util.sort_cols(calc.inner_summary().drop(["type", "r_squared_adj"], axis=1)).sort_index().astype(float))
    assert math.isclose(0.643594505232204, calc.goodness_of_fit())
    for scheme, fname in ((Scheme.PATH, "russa.outer_model_path.csv"), (Scheme.FACTORIAL, "russa.outer_model_factorial.csv")):
        exp = util.sort_cols(pd.read_csv(os.path.join(ref, fname), index_col=0).filter(["weight", "loading", "communality", "redundancy"])).sort_index()
        np.testing.assert_allclose(exp, util.sort_cols(Plspm(russa, config, scheme, 100, 0.0000001).outer_model()).sort_index())
    # seminr / mobi: PATH scheme, mixed Mode A / B, tolerance 1e-8
    mobi = pd.read_csv(os.path.join(ref, "mobi.csv"), index_col=0)
    st = c.Structure()
    st.add_path(["Expectation", "Quality"], ["Loyalty"]); st.add_path(["Image"], ["Expectation"]); st.add_path(["Complaints"], ["Loyalty"])
    cfg = c.Config(st.path(), default_scale=Scale.NUM)
    cfg.add_lv_with_columns_named("Expectation", Mode.A, mobi, "CUEX")
    cfg.add_lv_with_columns_named("Quality", Mode.B, mobi, "PERQ")
    cfg.add_lv_with_columns_named("Loyalty", Mode.A, mobi, "CUSL")
    cfg.add_lv_with_columns_named("Image", Mode.A, mobi, "IMAG")
    cfg.add_lv_with_columns_named("Complaints", Mode.A, mobi, "CUSCO")
    pls = Plspm(mobi, cfg, Scheme.PATH, 100, 0.00000001)
    exp_om = pd.read_csv(os.path.join(ref, "seminr-mobi-basic-outer-model.csv"), index_col=0)
    np.testing.assert_allclose(exp_om.sort_index(), pls.outer_model().drop(["communality", "redundancy"], axis=1).sort_index(), rtol=1e-5)
    exp_paths = pd.read_csv(os.path.join(ref, "seminr-mobi-basic-paths.csv"), index_col=0)
    np.testing.assert_allclose(exp_paths.sort_index().sort_index(axis=1), pls.path_coefficients().transpose().sort_index().sort_index(axis=1), rtol=1e-6)


def test_api_hoc_two_stage_reproduces_reference_seminr_test():
    """Mirror of reference tests/test_regression_seminr.py:49-74 (higher-order construct, two-stage approach, Scale.NUM)."""
    import os
    import plspm.config as c
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scale import Scale
    from plspm.scheme import Scheme
    ref = os.path.join(GOLDEN, "ref_data")
    mobi = pd.read_csv(os.path.join(ref, "mobi.csv"), index_col=0)
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
    pls = Plspm(mobi, config, Scheme.PATH, 100, 0.00000001)
    expected = pd.read_csv(os.path.join(ref, "seminr-mobi-hoc-ts-outer-model.csv"), index_col=0)
    actual = pls.outer_model().drop(["communality", "redundancy"], axis=1)
    common = sorted(set(expected.index) & set(actual.index))
    assert "Image" in common and "Value" in common and "PERQ4" in common
    np.testing.assert_allclose(expected.loc[common].sort_index(axis=1), actual.loc[common].sort_index(axis=1), rtol=1e-4)
    paths = pd.read_csv(os.path.join(ref, "seminr-mobi-hoc-ts-paths.csv"), index_col=0).transpose()
    np.testing.assert_allclose(paths.sort_index().sort_index(axis=1), pls.path_coefficients().sort_index().sort_index(axis=1), rtol=1e-6)
    assert list(pls.scores().columns) == list(structure.path()) if False else pls.scores().shape == (250, 5)


@pytest.mark.parametrize("B", [37, 5000, 20000])
def test_device_bootstrap_summary_matches_host_statistics(B):
    """plspm_bootstrap_summary (LDS sort up to 16k replicates, global scratch beyond) vs the NumPy / pandas definition."""
    from plspm.bootstrap import _create_summary
    X, blocks, _ = satisfaction_oracle_inputs()
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "centroid", False, max_iter=100)
    nm = native_model(model)
    nm.upload(X, model.mv_order.astype(np.int32))
    rows, status, _ = nm.bootstrap(B, seed=3)
    original = np.linspace(-1.0, 2.0, nm.row_width)
    table, used = nm.summary(B, original)
    ok = rows[status == 0]
    assert used == ok.shape[0] and used >= B - 5
    host = _create_summary(pd.DataFrame(ok), pd.Series(original)).values
    assert_close(table, host, 1e-11, 1e-13)
    assert_close(table, orc.summary(ok, original), 1e-11, 1e-13)


@pytest.mark.parametrize("B", [100, 5000, 9000])
def test_summary_of_device_records_equals_summary_of_the_same_records_stored_from_the_host(B):
    """plspm_bootstrap_summary on the records the bootstrap left on the handle, then on the same records fetched and stored back
    (plspm_bootstrap_store): identical tables; other records of the same count afterwards give their own table."""
    C = orc.satisfaction_C()
    X, blocks = orc.synth(600, C, 4, seed=3)
    model = orc.Model(blocks, C, "AAAAAA", "path", True)
    nm = native_model(model)
    nm.upload(X)
    nm.bootstrap_device(B, seed=5)
    original = np.linspace(-1.0, 2.0, nm.row_width)
    first, used = nm.summary(B, original)
    rows, status, iters = nm.fetch(0, B)
    from plspm import parallel
    nm.store(parallel.join_records(rows, status, iters))
    again, used2 = nm.summary(B, original)
    assert used == used2 == int((status == 0).sum())
    assert np.array_equal(first, again)
    assert_close(first, orc.summary(rows[status == 0], original), 1e-11, 1e-13)
    other = rows[::-1].copy()
    nm.store(parallel.join_records(other, status[::-1].copy(), iters[::-1].copy()))
    t3, _ = nm.summary(B, original)
    assert_close(t3, orc.summary(other[status[::-1] == 0], original), 1e-11, 1e-13)


def test_api_metric_missing_values_match_reference_golden():
    """Mean imputation + all-block-missing row drop (reference config.py:273-285,300) through the drop-in API."""
    import plspm.config as c
    from plspm.mode import Mode
    from plspm.plspm import Plspm
    from plspm.scheme import Scheme
    g = load("g10_metric_missing")
    _, blocks, cols = satisfaction_oracle_inputs()
    frame = pd.DataFrame(g["data"], columns=cols)
    structure = c.Structure()
    structure.add_path(["IMAG"], ["EXPE", "SAT", "LOY"]); structure.add_path(["EXPE"], ["QUAL", "VAL", "SAT"])
    structure.add_path(["QUAL"], ["VAL", "SAT"]); structure.add_path(["VAL"], ["SAT"]); structure.add_path(["SAT"], ["LOY"])
    for modes_name, modes in (("A", "AAAAAA"), ("M", "ABABAB")):
        for scheme_name, scheme in (("centroid", Scheme.CENTROID), ("path", Scheme.PATH)):
            for scaled in (False, True):
                config = c.Config(structure.path(), scaled=scaled)
                for lv in SAT_ADD_ORDER:
                    mode = Mode.A if modes[orc.SAT_LVS.index(lv)] == "A" else Mode.B
                    config.add_lv_with_columns_named(lv, mode, frame, SAT_PREFIX[lv])
                calc = Plspm(frame, config, scheme)
                key = "%s_%s_%d" % (modes_name, scheme_name, int(scaled))
                assert calc.iterations() == int(g[key + "/iters"])
                assert calc.scores().shape == (249, 6)
                om = calc.outer_model()
                assert_close(om.loc[cols, "weight"].values, g[key + "/weights"], RTOL)
                assert_close(om.loc[cols, "loading"].values, g[key + "/loadings"], RTOL)
                assert_close(calc.path_coefficients().loc[orc.SAT_LVS, orc.SAT_LVS].values, g[key + "/path_coef"], RTOL, ATOL)
                assert_close(calc.scores().loc[:, orc.SAT_LVS].values, g[key + "/scores"], 1e-7, 1e-9)
                uni = calc.unidimensionality()
                for lv in orc.SAT_LVS:                      # blocks with a missing value report NaN (unidimensionality.py:39)
                    has_nan = frame[[col for col in cols if col.startswith(SAT_PREFIX[lv])]].iloc[[i for i in range(250) if i != 7]].isnull().values.any()
                    assert bool(np.isnan(uni.loc[lv, "eig_1st"])) == bool(has_nan)


@pytest.mark.parametrize("L,per,n", [(30, 9, 900), (64, 15, 1500)])
def test_large_p_block_gram(L, per, n):
    """P = 270 (T = 18, partial last super-block) and P = 960 (T = 62): run-time super-block Gram + global-memory solver
    workspace, single fit vs the oracle and bootstrap rows vs the oracle on mirrored indices."""
    from plspm import _native
    C = orc.chain_C(L)
    X, blocks = orc.synth(n, C, per, seed=L)
    model = orc.Model(blocks, C, "A" * L, "centroid", True)
    nm, g = gpu_fit(X, model)
    r = orc.fit(X, model)
    check_fit(g, r, "P=%d" % (L * per))
    Xt = orc.treat_metric(X, True)
    assert_close(g["cov"], Xt.T @ Xt / n, 1e-10, 1e-13)
    rows, status, iters = nm.bootstrap(3, seed=4)
    assert np.all(status == 0)
    mine, its = orc.bootstrap_replicate(X, model, _native.bootstrap_indices(4, 2, n), orc.correction(n))
    assert its == iters[2]
    assert_close(rows[2], mine, RTOL, ATOL)


@pytest.mark.parametrize("B", [1, 2, 3, 7, 256, 257, 4097, 9000])
def test_device_summary_on_crafted_records(B):
    """summary_kernel on records written by the host (plspm_bootstrap_store): heavy ties around both quantiles, constant columns,
    negative values and signed zeros, values that differ only in their last bits, dropped replicates (status != 0 and the NaN status
    of a ragged shard's padding) -- the radix select stops early on single-value bins and must not on tied ones; the compaction
    loads 4,096 replicates per round.  Against the NumPy definition (orc.summary) on the rows that count."""
    X, blocks, _ = satisfaction_oracle_inputs()
    model = orc.Model(blocks, orc.satisfaction_C(), "AAAAAA", "centroid", False, max_iter=100)
    nm = native_model(model)
    nm.upload(X, model.mv_order.astype(np.int32))
    R = nm.row_width
    rng = np.random.default_rng(B)
    rows = rng.standard_normal((B, R))
    rows[:, 0] = 3.25                                              # constant column
    rows[:, 1] = rng.integers(0, 3, B).astype(float)               # three distinct values: ties at every rank
    rows[:, 2] = np.where(rng.random(B) < 0.5, 0.0, -0.0)          # signed zeros
    rows[:, 3] = -np.abs(rows[:, 3])                               # all negative
    rows[:, 4] = 1.0 + rng.integers(0, 4, B) * 2.0 ** -52          # neighbours in the last bit
    rows[:, 5] = np.round(rows[:, 5], 1)                           # many repeats
    status = np.zeros(B, dtype=np.int32)
    if B > 3:
        status[rng.random(B) < 0.1] = 1
        status[1] = 3
    iters = np.full(B, 4, dtype=np.int32)
    from plspm import parallel
    records = parallel.join_records(rows, status, iters)
    if B > 7:
        records[-2:, R] = np.nan                                   # padding records of a ragged shard
        status[-2:] = -1
    nm.store(records)
    original = np.linspace(-2.0, 2.0, R)
    table, used = nm.summary(B, original)
    ok = rows[status == 0]
    assert used == ok.shape[0]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                            # (NumPy on the std of a single value)
        want = orc.summary(ok, original)
    if ok.shape[0] < 2:
        assert np.all(np.isnan(table[:, 2])) and np.all(np.isnan(table[:, 5]))
        assert_close(table[:, [0, 1, 3, 4]], want[:, [0, 1, 3, 4]], 1e-12, 1e-14)
    else:
        finite = np.isfinite(want)
        finite[4, 2] = finite[4, 5] = False        # std of values that differ in their last bit is rounding noise in NumPy and here alike
        finite[0, 2] = finite[0, 5] = False        # (constant column: std 0 or an ulp of cancellation, t = +-inf or huge)
        assert np.all(np.isfinite(table[finite]))
        assert_close(table[finite], want[finite], 1e-11, 1e-13)
        assert table[4, 2] < 1e-15 and table[0, 2] < 1e-15


@pytest.mark.gpu
def test_nonmetric_numeric_bootstrap_beyond_65535_rows_on_the_int8_route():
    """Scale.NUM model, 70,000 rows: the digit-plane Gram with the dense stop-rule pass on its int8 counts against the fp64 route
    (global-histogram row lists, gathering pass): identical iteration counts, records to 1e-9; one replicate against the oracle."""
    from plspm import _native
    C = orc.chain_C(3)
    X, blocks = orc.synth(70000, C, 4, seed=43)
    boff = np.concatenate(([0], np.cumsum([len(b) for b in blocks]))).astype(np.int32)
    nm = _native.NativeModel(boff, C.astype(np.uint8), np.zeros(3, dtype=np.int32), 1, True, 100, 1e-6, 0, nonmetric=True)
    nm.upload(X)
    i8 = nm.bootstrap(40, seed=3)
    assert nm.get_option("last_gram_path") == 2 and np.all(i8[1] == 0)
    nm.set_option("gram_path", 1)
    f64 = nm.bootstrap(40, seed=3)
    assert nm.get_option("last_gram_path") == 1
    assert np.array_equal(i8[2], f64[2])
    np.testing.assert_allclose(i8[0], f64[0], rtol=1e-9, atol=1e-12)
    # round 4: EXPLICIT index lists beyond 65,535 rows stay on the digit planes as well (windowed 16-bit histograms, the stop-rule passes on
    # the int8 counts): fed the indices of the Philox stream, the records are the Philox route's bit for bit
    nm.set_option("gram_path", 0)
    idx = np.stack([_native.bootstrap_indices(3, r, 70000) for r in range(12)]).astype(np.int32)
    by_idx = nm.bootstrap(12, idx=idx)
    assert nm.get_option("last_gram_path") == 2
    assert np.array_equal(by_idx[0], i8[0][:12]) and np.array_equal(by_idx[2], i8[2][:12]) and np.all(by_idx[1] == 0)
    # ... and a list that carries a multiplicity above 127 falls back to the fp64 route for its chunk without changing a count
    heavy = idx.copy()
    heavy[0, :200] = 5
    hv = nm.bootstrap(12, idx=heavy)
    nm.set_option("gram_path", 1)
    hv64 = nm.bootstrap(12, idx=heavy)
    assert np.array_equal(hv[2], hv64[2]) and np.all(hv[1] == 0)
    np.testing.assert_allclose(hv[0], hv64[0], rtol=1e-9, atol=1e-12)
